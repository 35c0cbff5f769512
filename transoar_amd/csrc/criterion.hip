// Fused set criterion of the training step for gfx950 (include/transoar_criterion.h).
//
// The problem is tiny -- N x O x R = 2 x 20 x 27 queries, a dozen flops each -- and the host-side mirror spends ~270 launches of
// ~5 us on it per step (matcher geometry, one assignment per decoder layer, L1 / GIoU / BCE and their autograd graph).  Here it is
// ONE workgroup: a wave owns a (sample, class) row, a lane one query of the class; row maxima / minima / arg-minima are wave
// reductions; the gradients of L1, 1 - GIoU and the class loss with respect to the final layer's boxes and logits are produced in
// the same pass (closed forms of the formulas in transoar/utils/bboxes.py:6-43, the clamp / min / max sub-gradients as autograd
// takes them: the side that is active) and only scaled by the loss weights in the backward launch.
// Latency-bound by construction (one workgroup; a row is a chain of dependent loads and transcendentals): 16 waves take the
// flagship's 40 rows in three rounds.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/transoar_criterion.h"

namespace {

constexpr int kThreads = 1024, kWaves = kThreads / 64, kMaxAcc = 2 * TRANSOAR_CRIT_MAX_LAYERS + 1;      // forward: one workgroup of 16 waves
constexpr int kBwdThreads = 256;

struct LogitPtrs { const void* p[TRANSOAR_CRIT_MAX_LAYERS]; };

// position of decoder output l's (bbox, giou, cls) triple in the loss vector: the order of the criterion's dict (criterion.py:
// bbox, giou, cls, segce, segdice of the final output, then bbox_i, giou_i, cls_i of every auxiliary one)
__host__ __device__ __forceinline__ int loss_slot(int l) { return l == 0 ? 0 : 5 + 3 * (l - 1); }

template <typename T> __device__ __forceinline__ float load_f(const void* p, long i);
template <> __device__ __forceinline__ float load_f<float>(const void* p, long i) { return static_cast<const float*>(p)[i]; }
template <> __device__ __forceinline__ float load_f<unsigned short>(const void* p, long i) {
  return __uint_as_float(static_cast<unsigned>(static_cast<const unsigned short*>(p)[i]) << 16);
}
__device__ __forceinline__ unsigned short bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<unsigned short>((u >> 16) | 0x40u);      // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<unsigned short>(u >> 16);
}
template <typename T> __device__ __forceinline__ void store_f(void* p, long i, float v);
template <> __device__ __forceinline__ void store_f<float>(void* p, long i, float v) { static_cast<float*>(p)[i] = v; }
template <> __device__ __forceinline__ void store_f<unsigned short>(void* p, long i, float v) { static_cast<unsigned short*>(p)[i] = bf16_rne(v); }

__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// GIoU of box a = (c, s) (centre / size, already clamped) against b given as corners (bboxes.py:6-43, the operations in the
// mirror's order); with GRAD also d(giou) / d(c_k), d(giou) / d(s_k).
template <bool GRAD>
__device__ __forceinline__ float giou_cs(const float (&c)[3], const float (&s)[3], const float (&blo)[3], const float (&bhi)[3],
                                         float (&dc)[3], float (&ds)[3]) {
#pragma clang fp contract(off)
  float alo[3], ahi[3], ik[3], hk[3], ea[3], eb[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    alo[k] = c[k] - 0.5f * s[k];
    ahi[k] = c[k] + 0.5f * s[k];
    ik[k] = fmaxf(fminf(ahi[k], bhi[k]) - fmaxf(alo[k], blo[k]), 0.f);
    hk[k] = fmaxf(fmaxf(ahi[k], bhi[k]) - fminf(alo[k], blo[k]), 0.f);
    ea[k] = ahi[k] - alo[k];
    eb[k] = bhi[k] - blo[k];
  }
  const float inter = ik[0] * ik[1] * ik[2];
  const float va = ea[0] * ea[1] * ea[2], vb = eb[0] * eb[1] * eb[2];
  const float uni = va + vb - inter;
  const float iou = inter / uni;
  const float hull = hk[0] * hk[1] * hk[2];
  const float giou = iou - (hull - uni) / hull;
  if constexpr (GRAD) {
    // giou = inter / uni - 1 + uni / hull
    const float g_inter = 1.f / uni, g_uni = -inter / (uni * uni) + 1.f / hull, g_hull = -uni / (hull * hull);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
      const float oi = ik[k1] * ik[k2], oh = hk[k1] * hk[k2], oa = ea[k1] * ea[k2];
      const float iarg = fminf(ahi[k], bhi[k]) - fmaxf(alo[k], blo[k]), harg = fmaxf(ahi[k], bhi[k]) - fminf(alo[k], blo[k]);
      // partials of the axis' intersection / hull extent / own extent with respect to (a_lo, a_hi)
      const float i_hi = (iarg >= 0.f && ahi[k] <= bhi[k]) ? 1.f : 0.f, i_lo = (iarg >= 0.f && alo[k] >= blo[k]) ? -1.f : 0.f;
      const float h_hi = (harg >= 0.f && ahi[k] >= bhi[k]) ? 1.f : 0.f, h_lo = (harg >= 0.f && alo[k] <= blo[k]) ? -1.f : 0.f;
      // d(inter), d(vol_a), d(hull) per corner; d(uni) = d(vol_a) - d(inter)
      const float dI_hi = oi * i_hi, dI_lo = oi * i_lo, dA_hi = oa, dA_lo = -oa, dH_hi = oh * h_hi, dH_lo = oh * h_lo;
      const float d_hi = g_inter * dI_hi + g_uni * (dA_hi - dI_hi) + g_hull * dH_hi;
      const float d_lo = g_inter * dI_lo + g_uni * (dA_lo - dI_lo) + g_hull * dH_lo;
      dc[k] = d_lo + d_hi;
      ds[k] = 0.5f * (d_hi - d_lo);
    }
  }
  return giou;
}

template <typename LT, typename BT>
__global__ __launch_bounds__(kThreads) void set_criterion_fwd(
    LogitPtrs logits, int layers, const void* __restrict__ boxes, const float* __restrict__ anchors, const float* __restrict__ tgt,
    const unsigned char* __restrict__ present, const float* __restrict__ nb_dev, float nb_host, const float* __restrict__ np_dev,
    float cost_class, float cost_bbox, float cost_giou, int N, int O, int R, float* __restrict__ losses, float* __restrict__ d_l1,
    float* __restrict__ d_giou, float* __restrict__ d_cls, unsigned char* __restrict__ hit) {
#pragma clang fp contract(off)
  __shared__ int s_present;
  __shared__ double s_acc[kWaves][kMaxAcc];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rows = N * O, Q = O * R;
  if (tid == 0) s_present = 0;
  __syncthreads();
  {
    int c = 0;
    for (int i = tid; i < rows; i += kThreads) c += present[i] ? 1 : 0;
    if (c) atomicAdd(&s_present, c);
  }
  __syncthreads();
  const float num_boxes = nb_dev ? *nb_dev : nb_host;
  const float n_valid = np_dev ? *np_dev * static_cast<float>(R) : static_cast<float>(s_present * R);
  const float inv_nb = 1.f / num_boxes;
  double acc[kMaxAcc];
#pragma unroll
  for (int i = 0; i < kMaxAcc; ++i) acc[i] = 0.0;
  const bool active = lane < R;
  const float inf = __builtin_inff();
  for (int row = wave; row < rows; row += kWaves) {
    const int n = row / O, o = row - n * O;
    const bool pres = present[row] != 0;
    float t[6], blo[3], bhi[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] = tgt[static_cast<long>(row) * 6 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { blo[k] = t[k] - 0.5f * t[3 + k]; bhi[k] = t[k] + 0.5f * t[3 + k]; }
    const long item = static_cast<long>(n) * Q + o * R + lane;
    // ---- the anchor side of the matching (matcher.py:40-57): -GIoU and L1 of the class's anchors against its target
    float cg = 0.f, l1a = 0.f, dum_c[3], dum_s[3];
    if (active) {
      float a[6], ac[3], as[3];
#pragma unroll
      for (int k = 0; k < 6; ++k) a[k] = anchors[static_cast<long>(o * R + lane) * 6 + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) { ac[k] = fmaxf(a[k], 0.f); as[k] = fmaxf(a[3 + k], 0.f); }
      cg = -giou_cs<false>(ac, as, blo, bhi, dum_c, dum_s);
#pragma unroll
      for (int k = 0; k < 6; ++k) l1a += fabsf(a[k] - t[k]);
    }
    const float hi = wave_max_f(active ? cg : -inf), lo = wave_min_f(active ? cg : inf);
    float soft = (cg - hi) / (lo - hi);
    soft = (soft != soft) ? soft : fmaxf(soft, 0.f);             // clamp(min = 0) keeps a NaN (0 / 0: one distinct GIoU in the row)
    if (!pres) soft = -1.f;
    // ---- the final layer's terms of this query (criterion.py:52-74)
    float l1f = 0.f, omg = 0.f;
    if (active) {
      float p[6], pc[3], ps[3], dc[3], ds[3];
#pragma unroll
      for (int k = 0; k < 6; ++k) p[k] = load_f<BT>(boxes, item * 6 + k);
#pragma unroll
      for (int k = 0; k < 3; ++k) { pc[k] = fmaxf(p[k], 0.f); ps[k] = fmaxf(p[3 + k], 0.f); }
      omg = 1.f - giou_cs<true>(pc, ps, blo, bhi, dc, ds);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float d = p[k] - t[k];
        l1f += fabsf(d);
        d_l1[item * 6 + k] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * inv_nb;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        d_giou[item * 6 + k] = (p[k] >= 0.f ? -dc[k] : 0.f) * inv_nb;
        d_giou[item * 6 + 3 + k] = (p[3 + k] >= 0.f ? -ds[k] : 0.f) * inv_nb;
      }
      // ---- class loss of the final logits against the soft labels (criterion.py:40-50)
      const float x = load_f<LT>(logits.p[0], item), y = (soft != soft) ? soft : fmaxf(soft, 0.f);
      const bool valid = soft != -1.f;
      const float per = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
      if (valid) acc[2 * layers] += static_cast<double>(per);
      d_cls[item] = valid ? (1.f / (1.f + expf(-x)) - y) / n_valid : 0.f;
    }
    // ---- one assignment per decoder output (matcher.py:59-65, criterion.py:114-123): the cheapest query of the class
    for (int l = 0; l < layers; ++l) {
      float key = inf;
      if (active) {
        const float x = load_f<LT>(logits.p[l], item);
        const float prob = 1.f / (1.f + expf(-x));
        float cost = cost_class * (-prob) + cost_giou * cg;
        if (cost_bbox != 0.f) cost = cost + cost_bbox * l1a;
        key = (cost != cost) ? inf : cost;                          // a NaN cost never wins (topk orders NaN last)
      }
      const float best = wave_min_f(key);
      const unsigned long long who = __ballot(active && key == best);
      const int winner = who ? __ffsll(static_cast<long long>(who)) - 1 : -1;
      const bool h = pres && active && lane == winner;
      if (active) hit[(static_cast<long>(l) * N + n) * Q + o * R + lane] = h ? 1 : 0;
      if (h) { acc[2 * l] += static_cast<double>(l1f); acc[2 * l + 1] += static_cast<double>(omg); }
    }
  }
  const int n_acc = 2 * layers + 1;
  for (int i = 0; i < n_acc; ++i) {
    const double v = wave_sum_d(acc[i]);
    if (lane == 0) s_acc[wave][i] = v;
  }
  __syncthreads();
  if (tid < n_acc) {
    double v = 0.0;
    for (int w = 0; w < kWaves; ++w) v += s_acc[w][tid];
    if (tid == 2 * layers) {                       // the class loss: the final output's slot and every auxiliary output's
      const float c = static_cast<float>(v) / n_valid;
      losses[2] = c;
      for (int l = 1; l < layers; ++l) losses[loss_slot(l) + 2] = c;
      losses[3] = 0.f;                             // the segmentation proxy losses: not in this form
      losses[4] = 0.f;
    } else {
      losses[loss_slot(tid >> 1) + (tid & 1)] = static_cast<float>(v) / num_boxes;
    }
  }
}

template <typename LT, typename BT>
__global__ __launch_bounds__(kBwdThreads) void set_criterion_bwd(const float* __restrict__ g, int layers, const float* __restrict__ d_l1,
                                                              const float* __restrict__ d_giou, const float* __restrict__ d_cls,
                                                              const unsigned char* __restrict__ hit, long items,
                                                              void* __restrict__ grad_boxes, void* __restrict__ grad_logits) {
  const long i = static_cast<long>(blockIdx.x) * kBwdThreads + threadIdx.x;
  if (i >= items) return;
  float a = 0.f, b = 0.f, c = 0.f;
  bool any = false;
  for (int l = 0; l < layers; ++l) {
    const int s = loss_slot(l);
    c += g[s + 2];
    if (hit[static_cast<long>(l) * items + i]) { a += g[s]; b += g[s + 1]; any = true; }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) store_f<BT>(grad_boxes, i * 6 + k, any ? a * d_l1[i * 6 + k] + b * d_giou[i * 6 + k] : 0.f);
  store_f<LT>(grad_logits, i, c * d_cls[i]);
}

bool dtype_ok(int d) { return d == TRANSOAR_CRIT_F32 || d == TRANSOAR_CRIT_BF16; }

}  // namespace

extern "C" int transoar_set_criterion_forward(const void* const* logits, int layers, int logits_dtype, const void* boxes, int boxes_dtype,
                                              const float* anchors, const float* tgt_boxes, const unsigned char* present,
                                              const float* num_boxes_dev, float num_boxes_host, const float* n_present_dev,
                                              float cost_class, float cost_bbox, float cost_giou, int N, int O, int R, float* losses,
                                              float* d_l1, float* d_giou, float* d_cls, unsigned char* hit, void* hip_stream) {
  if (!logits || !boxes || !anchors || !tgt_boxes || !present || !losses || !d_l1 || !d_giou || !d_cls || !hit) return TRANSOAR_CRIT_ERR_NULL;
  if (layers < 1 || layers > TRANSOAR_CRIT_MAX_LAYERS || N < 1 || O < 1 || R < 2 || R > TRANSOAR_CRIT_MAX_R ||
      static_cast<long>(N) * O * R > (1L << 24))
    return TRANSOAR_CRIT_ERR_DIM;
  if (!dtype_ok(logits_dtype) || !dtype_ok(boxes_dtype)) return TRANSOAR_CRIT_ERR_DTYPE;
  LogitPtrs lp{};
  for (int l = 0; l < layers; ++l) {
    if (!logits[l]) return TRANSOAR_CRIT_ERR_NULL;
    lp.p[l] = logits[l];
  }
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
#define TRANSOAR_CRIT_LAUNCH(LT, BT)                                                                                            \
  hipLaunchKernelGGL((set_criterion_fwd<LT, BT>), dim3(1), dim3(kThreads), 0, st, lp, layers, boxes, anchors, tgt_boxes, present, \
                     num_boxes_dev, num_boxes_host, n_present_dev, cost_class, cost_bbox, cost_giou, N, O, R, losses, d_l1, d_giou,  \
                     d_cls, hit)
  if (logits_dtype == TRANSOAR_CRIT_F32 && boxes_dtype == TRANSOAR_CRIT_F32) TRANSOAR_CRIT_LAUNCH(float, float);
  else if (logits_dtype == TRANSOAR_CRIT_BF16 && boxes_dtype == TRANSOAR_CRIT_F32) TRANSOAR_CRIT_LAUNCH(unsigned short, float);
  else if (logits_dtype == TRANSOAR_CRIT_F32 && boxes_dtype == TRANSOAR_CRIT_BF16) TRANSOAR_CRIT_LAUNCH(float, unsigned short);
  else TRANSOAR_CRIT_LAUNCH(unsigned short, unsigned short);
#undef TRANSOAR_CRIT_LAUNCH
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_set_criterion_backward(const float* g, int layers, const float* d_l1, const float* d_giou, const float* d_cls,
                                               const unsigned char* hit, int N, int O, int R, void* grad_boxes, int boxes_dtype,
                                               void* grad_logits, int logits_dtype, void* hip_stream) {
  if (!g || !d_l1 || !d_giou || !d_cls || !hit || !grad_boxes || !grad_logits) return TRANSOAR_CRIT_ERR_NULL;
  if (layers < 1 || layers > TRANSOAR_CRIT_MAX_LAYERS || N < 1 || O < 1 || R < 2 || R > TRANSOAR_CRIT_MAX_R) return TRANSOAR_CRIT_ERR_DIM;
  if (!dtype_ok(logits_dtype) || !dtype_ok(boxes_dtype)) return TRANSOAR_CRIT_ERR_DTYPE;
  const long items = static_cast<long>(N) * O * R;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const dim3 grid(static_cast<unsigned>((items + kBwdThreads - 1) / kBwdThreads));
#define TRANSOAR_CRIT_LAUNCH(LT, BT) \
  hipLaunchKernelGGL((set_criterion_bwd<LT, BT>), grid, dim3(kBwdThreads), 0, st, g, layers, d_l1, d_giou, d_cls, hit, items, grad_boxes, grad_logits)
  if (logits_dtype == TRANSOAR_CRIT_F32 && boxes_dtype == TRANSOAR_CRIT_F32) TRANSOAR_CRIT_LAUNCH(float, float);
  else if (logits_dtype == TRANSOAR_CRIT_BF16 && boxes_dtype == TRANSOAR_CRIT_F32) TRANSOAR_CRIT_LAUNCH(unsigned short, float);
  else if (logits_dtype == TRANSOAR_CRIT_F32 && boxes_dtype == TRANSOAR_CRIT_BF16) TRANSOAR_CRIT_LAUNCH(float, unsigned short);
  else TRANSOAR_CRIT_LAUNCH(unsigned short, unsigned short);
#undef TRANSOAR_CRIT_LAUNCH
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_criterion_abi_version(void) { return 1; }
