// Fused residual-add + LayerNorm (+ casts, + positional add) over the token
// stream of the refine block; C ABI in include/transoar_tokens.h.  gfx950 only.
//
// One wave per token row: cols = 128*K columns, lane l holds columns
// i*128 + 2l, +1 (i < K) -> every load/store of a row is one contiguous
// 512-byte (fp32) or 256-byte (bf16) wave access.  Statistics are two-pass in
// registers (mean, then centred second moment), fp32.  HBM-bound by design:
// forward moves 18 B/element (fp32 stream) where the unfused chain moves 48.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/transoar_tokens.h"

namespace {

constexpr int kWaves = 4;                       // rows per workgroup
constexpr int kPersistentWaves = 256 * 4 * kWaves;   // backward: 4 workgroups per CU (16 waves: the row walk is latency-bound)

__device__ __forceinline__ float bf16_lo(unsigned int u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned int f32_to_bf16_bits(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ unsigned int pack_bf16(float a, float b) {
  return f32_to_bf16_bits(a) | (f32_to_bf16_bits(b) << 16);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Where a dropout keep-mask comes from: bytes in memory (1 = keep), or a counter-based hash of the element index
// and a per-call seed drawn from torch's generator (nothing is written or read: 240 MB per FFN hidden tensor
// less, each way).  Two elements per call: low byte / bit 8 of the result, like two mask bytes.
struct KeepSrc {
  const unsigned short* bytes;
  const int* seed;
  unsigned thr16;        // keep when a uniform 16-bit value < thr16
  __device__ __forceinline__ bool active() const { return bytes != nullptr || seed != nullptr; }
};
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned short keep_pair(const KeepSrc& ks, unsigned seed, long pair) {
  if (ks.bytes != nullptr) return ks.bytes[pair];
  const unsigned h = hash32(static_cast<unsigned>(pair) * 0x9e3779b9u + seed);
  return static_cast<unsigned short>(((h & 0xffffu) < ks.thr16 ? 1u : 0u) | ((h >> 16) < ks.thr16 ? 0x100u : 0u));
}

// row of the residual stream (+ branch) into registers
// (the branch r may still need its dropout: keep[e] != 0 -> r[e] * scale, else 0)
template <int K, bool XBF>
__device__ __forceinline__ void load_sum(const void* x, const unsigned int* r, const KeepSrc& keep, float scale,
                                         long row, int cols, int lane, float (&v)[2 * K],
                                         unsigned short (&mask)[K]) {
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const long e = row * cols + i * 128 + 2 * lane;
    float a, b;
    if (XBF) {
      const unsigned int u = static_cast<const unsigned int*>(x)[e >> 1];
      a = bf16_lo(u); b = bf16_hi(u);
    } else {
      const float2 f = *reinterpret_cast<const float2*>(static_cast<const float*>(x) + e);
      a = f.x; b = f.y;
    }
    if (r != nullptr) {
      const unsigned int u = r[e >> 1];
      float ra = bf16_lo(u), rb = bf16_hi(u);
      mask[i] = 0x0101;
      if (keep.active()) {
        const unsigned short m = keep_pair(keep, keep.seed ? static_cast<unsigned>(*keep.seed) : 0u, e >> 1);   // two mask bytes
        mask[i] = m;
        ra = (m & 0xffu) ? ra * scale : 0.f;
        rb = (m >> 8) ? rb * scale : 0.f;
      }
      a += ra; b += rb;
    }
    v[2 * i] = a; v[2 * i + 1] = b;
  }
}

__device__ __forceinline__ int level_of(const int* level_start, int L, int s) {
  int l = 0;
  for (int t = 1; t < L; ++t) l += (s >= level_start[t]) ? 1 : 0;
  return l;
}

template <int K, bool XBF>
__global__ __launch_bounds__(64 * kWaves) void add_ln_fwd(
    const void* __restrict__ x, const unsigned int* __restrict__ r, const float* __restrict__ weight,
    const float* __restrict__ bias, float eps, const float* __restrict__ pos_sine,
    const float* __restrict__ level_embed, const int* __restrict__ level_start, int L, long S,
    float* __restrict__ y32, unsigned int* __restrict__ y16, unsigned int* __restrict__ q16,
    float* __restrict__ mean_rstd, long rows, KeepSrc keep, float scale) {
  constexpr int cols = 128 * K;
  const int lane = threadIdx.x & 63;
  const long row = static_cast<long>(blockIdx.x) * kWaves + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[2 * K];
  unsigned short mask[K];
  load_sum<K, XBF>(x, r, keep, scale, row, cols, lane, v, mask);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2 * K; ++i) s += v[i];
  const float mean = wave_sum(s) * (1.0f / cols);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 2 * K; ++i) { v[i] -= mean; q += v[i] * v[i]; }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / cols) + eps);
  if (lane == 0) *reinterpret_cast<float2*>(mean_rstd + 2 * row) = float2{mean, rstd};
  const long srow = row % S;
  const int lvl = q16 != nullptr ? level_of(level_start, L, static_cast<int>(srow)) : 0;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int c = i * 128 + 2 * lane;
    const float2 w = *reinterpret_cast<const float2*>(weight + c), b = *reinterpret_cast<const float2*>(bias + c);
    const float y0 = v[2 * i] * rstd * w.x + b.x, y1 = v[2 * i + 1] * rstd * w.y + b.y;
    const long e = row * cols + c;
    *reinterpret_cast<float2*>(y32 + e) = float2{y0, y1};
    y16[e >> 1] = pack_bf16(y0, y1);
    if (q16 != nullptr) {
      const float2 ps = *reinterpret_cast<const float2*>(pos_sine + srow * cols + c);
      const float2 le = *reinterpret_cast<const float2*>(level_embed + lvl * cols + c);
      q16[e >> 1] = pack_bf16(y0 + (ps.x + le.x), y1 + (ps.y + le.y));
    }
  }
}

template <int K, bool XBF>
__global__ __launch_bounds__(64 * kWaves) void add_ln_bwd(
    const float* __restrict__ g32, const unsigned int* __restrict__ g16, const unsigned int* __restrict__ gq16,
    const void* __restrict__ x, const unsigned int* __restrict__ r, const float* __restrict__ weight,
    const float* __restrict__ mean_rstd, const int* __restrict__ level_start, int L, long S,
    void* __restrict__ gx, unsigned int* __restrict__ gr16, float* __restrict__ partials, long rows,
    KeepSrc keep, float scale) {
  constexpr int cols = 128 * K;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * kWaves + (threadIdx.x >> 6);
  float w[2 * K], dw[2 * K], db[2 * K];
  float dle[TRANSOAR_TOK_MAX_LEVELS][2 * K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const float2 f = *reinterpret_cast<const float2*>(weight + i * 128 + 2 * lane);
    w[2 * i] = f.x; w[2 * i + 1] = f.y;
  }
#pragma unroll
  for (int i = 0; i < 2 * K; ++i) dw[i] = db[i] = 0.f;
#pragma unroll
  for (int l = 0; l < TRANSOAR_TOK_MAX_LEVELS; ++l)
#pragma unroll
    for (int i = 0; i < 2 * K; ++i) dle[l][i] = 0.f;

  for (long row = wave; row < rows; row += kPersistentWaves) {
    float v[2 * K], g[2 * K];
    unsigned short mask[K];
    load_sum<K, XBF>(x, r, keep, scale, row, cols, lane, v, mask);
    const float2 mr = *reinterpret_cast<const float2*>(mean_rstd + 2 * row);
    const int lvl = gq16 != nullptr ? level_of(level_start, L, static_cast<int>(row % S)) : 0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const long e = row * cols + i * 128 + 2 * lane;
      float a = 0.f, b = 0.f;
      if (g32 != nullptr) {
        const float2 f = *reinterpret_cast<const float2*>(g32 + e);
        a = f.x; b = f.y;
      }
      if (g16 != nullptr) {
        const unsigned int u = g16[e >> 1];
        a += bf16_lo(u); b += bf16_hi(u);
      }
      if (gq16 != nullptr) {
        const unsigned int u = gq16[e >> 1];
        const float qa = bf16_lo(u), qb = bf16_hi(u);
        a += qa; b += qb;
        // uniform level per row: a switch keeps dle[] in registers
#pragma unroll
        for (int l = 0; l < TRANSOAR_TOK_MAX_LEVELS; ++l)
          if (l == lvl) { dle[l][2 * i] += qa; dle[l][2 * i + 1] += qb; }
      }
      g[2 * i] = a; g[2 * i + 1] = b;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * K; ++i) {
      v[i] = (v[i] - mr.x) * mr.y;                 // xhat
      db[i] += g[i];
      dw[i] += g[i] * v[i];
      g[i] *= w[i];
      s1 += g[i];
      s2 += g[i] * v[i];
    }
    s1 = wave_sum(s1) * (1.0f / cols);
    s2 = wave_sum(s2) * (1.0f / cols);
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const long e = row * cols + i * 128 + 2 * lane;
      const float d0 = (g[2 * i] - s1 - v[2 * i] * s2) * mr.y, d1 = (g[2 * i + 1] - s1 - v[2 * i + 1] * s2) * mr.y;
      if (XBF) {
        static_cast<unsigned int*>(gx)[e >> 1] = pack_bf16(d0, d1);
      } else {
        *reinterpret_cast<float2*>(static_cast<float*>(gx) + e) = float2{d0, d1};
      }
      if (gr16 != nullptr) {                      // gradient of the (pre-dropout) branch
        float r0 = d0, r1 = d1;
        if (keep.active()) {
          r0 = (mask[i] & 0xffu) ? d0 * scale : 0.f;
          r1 = (mask[i] >> 8) ? d1 * scale : 0.f;
        }
        gr16[e >> 1] = pack_bf16(r0, r1);
      }
    }
  }
  float* out = partials + static_cast<long>(wave) * (2 + L) * cols;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int c = i * 128 + 2 * lane;
    *reinterpret_cast<float2*>(out + c) = float2{dw[2 * i], dw[2 * i + 1]};
    *reinterpret_cast<float2*>(out + cols + c) = float2{db[2 * i], db[2 * i + 1]};
#pragma unroll
    for (int l = 0; l < TRANSOAR_TOK_MAX_LEVELS; ++l)
      if (l < L) *reinterpret_cast<float2*>(out + (2 + l) * cols + c) = float2{dle[l][2 * i], dle[l][2 * i + 1]};
  }
}

// y = keep ? relu(h) * scale : 0 on bf16, 8 elements per thread
__global__ __launch_bounds__(256) void relu_dropout_fwd(const uint4* __restrict__ h, KeepSrc keep, float scale,
                                                        uint4* __restrict__ y, long n8) {
  const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n8) return;
  const uint4 u = h[i];
  const unsigned seed = keep.seed ? static_cast<unsigned>(*keep.seed) : 0u;
  const unsigned int in[4] = {u.x, u.y, u.z, u.w};
  unsigned int out[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned int mb = keep.active() ? keep_pair(keep, seed, 4 * i + j) : 0x0101u;
    const float a = fmaxf(bf16_lo(in[j]), 0.f), b = fmaxf(bf16_hi(in[j]), 0.f);
    out[j] = pack_bf16((mb & 0xffu) ? a * scale : 0.f, (mb & 0xff00u) ? b * scale : 0.f);
  }
  y[i] = uint4{out[0], out[1], out[2], out[3]};
}
// gh = y > 0 ? gy * scale : 0   (y > 0 <=> kept and h > 0)
__global__ __launch_bounds__(256) void relu_dropout_bwd(const uint4* __restrict__ gy, const uint4* __restrict__ y,
                                                        float scale, uint4* __restrict__ gh, long n8) {
  const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n8) return;
  const uint4 g = gy[i], o = y[i];
  const unsigned int gi[4] = {g.x, g.y, g.z, g.w}, oi[4] = {o.x, o.y, o.z, o.w};
  unsigned int out[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    out[j] = pack_bf16(bf16_lo(oi[j]) > 0.f ? bf16_lo(gi[j]) * scale : 0.f,
                       bf16_hi(oi[j]) > 0.f ? bf16_hi(gi[j]) * scale : 0.f);
  gh[i] = uint4{out[0], out[1], out[2], out[3]};
}

// ---------------------------------------------------------------------------
// Head of MSDeformAttn (ms_deform_attn.py:114-128 of the reference) on the stacked projection
// proj (T, M*G*3 + M*G) bf16 = [sampling offsets (M, L, P, 3) | attention logits (M, L*P)], G = L*P:
//     loc  (T, M, L, P, 3) fp32 = ref (T, L, 3) + bf16(off / bf16(W_l, H_l, D_l))
//     attn (T, M, G)      fp32 = softmax over G of the logits (fp32 arithmetic)
// -- the rounding points of the eager chain under autocast (bf16 division, fp32 add and softmax).  One thread
// per (token, head, level*point); the G logits of a head meet in LDS.  Backward: the gradient of proj in bf16.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float bf16_round(float f) { return __uint_as_float(f32_to_bf16_bits(f) << 16); }

__global__ __launch_bounds__(256) void sampling_head_fwd(const unsigned short* __restrict__ proj,
                                                         const float* __restrict__ ref, const long* __restrict__ shapes,
                                                         float* __restrict__ loc, float* __restrict__ attn, int M, int L,
                                                         int P, long n_elem, long ref_rows) {
  __shared__ float sh[256];
  const int G = L * P, cols = 4 * M * G;
  const long e = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool live = e < n_elem;
  const long ec = live ? e : n_elem - 1;
  const int g = static_cast<int>(ec % G);
  const long tm = ec / G;
  const int m = static_cast<int>(tm % M);
  const long t = tm / M;
  const unsigned short* row = proj + t * cols;
  const float logit = __uint_as_float(static_cast<unsigned>(row[3 * M * G + m * G + g]) << 16);
  sh[threadIdx.x] = logit;
  __syncthreads();
  float* grp = sh + (threadIdx.x - g);
  float mx = grp[0];
  for (int i = 1; i < G; ++i) mx = fmaxf(mx, grp[i]);
  const float ex = __expf(logit - mx);           // one exponential per thread; the group's sum through LDS again
  __syncthreads();
  sh[threadIdx.x] = ex;
  __syncthreads();
  float sum = 0.f;
  for (int i = 0; i < G; ++i) sum += grp[i];
  if (!live) return;
  attn[e] = ex / sum;
  const int l = g / P;
  const unsigned short* o = row + (m * G + g) * 3;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float size = bf16_round(static_cast<float>(shapes[l * 3 + 2 - k]));       // (W, H, D) for (x, y, z)
    const float off = __uint_as_float(static_cast<unsigned>(o[k]) << 16);
    loc[e * 3 + k] = ref[((t % ref_rows) * L + l) * 3 + k] + bf16_round(off / size);
  }
}

__global__ __launch_bounds__(256) void sampling_head_bwd(const float* __restrict__ g_loc, const float* __restrict__ g_attn,
                                                         const float* __restrict__ attn, const long* __restrict__ shapes,
                                                         unsigned short* __restrict__ g_proj, int M, int L, int P,
                                                         long n_elem) {
  __shared__ float sh[256];
  const int G = L * P, cols = 4 * M * G;
  const long e = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool live = e < n_elem;
  const long ec = live ? e : n_elem - 1;
  const int g = static_cast<int>(ec % G);
  const long tm = ec / G;
  const int m = static_cast<int>(tm % M);
  const long t = tm / M;
  const float a = attn[ec], ga = g_attn[ec];
  sh[threadIdx.x] = a * ga;
  __syncthreads();
  const float* grp = sh + (threadIdx.x - g);
  float dot = 0.f;
  for (int i = 0; i < G; ++i) dot += grp[i];
  if (!live) return;
  unsigned short* row = g_proj + t * cols;
  row[3 * M * G + m * G + g] = static_cast<unsigned short>(f32_to_bf16_bits(a * (ga - dot)));
  const int l = g / P;
  unsigned short* o = row + (m * G + g) * 3;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float size = bf16_round(static_cast<float>(shapes[l * 3 + 2 - k]));
    o[k] = static_cast<unsigned short>(f32_to_bf16_bits(bf16_round(g_loc[e * 3 + k]) / size));
  }
}

}  // namespace

static unsigned keep_threshold(float keep_prob) {
  const float t = keep_prob * 65536.0f + 0.5f;
  return t <= 0.f ? 0u : (t >= 65535.f ? 65535u : static_cast<unsigned>(t));
}

#define TOK_DISPATCH(K_, BODY) \
  switch (K_) {                \
    case 1: { constexpr int K = 1; BODY; break; } \
    case 2: { constexpr int K = 2; BODY; break; } \
    case 3: { constexpr int K = 3; BODY; break; } \
    case 4: { constexpr int K = 4; BODY; break; } \
    case 6: { constexpr int K = 6; BODY; break; } \
    case 8: { constexpr int K = 8; BODY; break; } \
    default: return TRANSOAR_TOK_ERR_DIM;         \
  }

extern "C" int transoar_add_layernorm_forward(const void* x, int x_is_bf16, const void* r, const float* weight,
                                              const float* bias, float eps, const float* pos_sine,
                                              const float* level_embed, const int* level_start, int L, long S,
                                              float* y32, void* y16, void* q16, float* mean_rstd, long rows,
                                              int cols, const unsigned char* keep, float keep_scale,
                                              const int* keep_seed, float keep_prob, void* hip_stream) {
  if (!x || !weight || !bias || !y32 || !y16 || !mean_rstd) return TRANSOAR_TOK_ERR_NULL;
  if (q16 && (!pos_sine || !level_embed || !level_start)) return TRANSOAR_TOK_ERR_NULL;
  if (rows <= 0 || cols <= 0 || cols % 128 || cols > 1024 || S <= 0) return TRANSOAR_TOK_ERR_DIM;
  if (L < 0 || L > TRANSOAR_TOK_MAX_LEVELS || (q16 && L == 0)) return TRANSOAR_TOK_ERR_LEVELS;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const dim3 grid(static_cast<unsigned>((rows + kWaves - 1) / kWaves)), block(64 * kWaves);
  auto rr = static_cast<const unsigned int*>(r);
  auto o16 = static_cast<unsigned int*>(y16);
  auto oq = static_cast<unsigned int*>(q16);
  const KeepSrc kp{reinterpret_cast<const unsigned short*>(keep), keep_seed, keep_threshold(keep_prob)};
  TOK_DISPATCH(cols / 128, {
    if (x_is_bf16)
      hipLaunchKernelGGL((add_ln_fwd<K, true>), grid, block, 0, st, x, rr, weight, bias, eps, pos_sine, level_embed,
                         level_start, L, S, y32, o16, oq, mean_rstd, rows, kp, keep_scale);
    else
      hipLaunchKernelGGL((add_ln_fwd<K, false>), grid, block, 0, st, x, rr, weight, bias, eps, pos_sine, level_embed,
                         level_start, L, S, y32, o16, oq, mean_rstd, rows, kp, keep_scale);
  });
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_add_layernorm_backward(const float* g32, const void* g16, const void* gq16, const void* x,
                                               int x_is_bf16, const void* r, const float* weight,
                                               const float* mean_rstd, const int* level_start, int L, long S,
                                               void* gx, void* gr16, float* partials, long rows, int cols,
                                               const unsigned char* keep, float keep_scale, const int* keep_seed,
                                               float keep_prob, void* hip_stream) {
  if (!x || !weight || !mean_rstd || !gx || !partials) return TRANSOAR_TOK_ERR_NULL;
  if (gq16 && !level_start) return TRANSOAR_TOK_ERR_NULL;
  if (rows <= 0 || cols <= 0 || cols % 128 || cols > 1024 || S <= 0) return TRANSOAR_TOK_ERR_DIM;
  if (L < 0 || L > TRANSOAR_TOK_MAX_LEVELS) return TRANSOAR_TOK_ERR_LEVELS;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const dim3 grid(kPersistentWaves / kWaves), block(64 * kWaves);
  auto a16 = static_cast<const unsigned int*>(g16);
  auto aq = static_cast<const unsigned int*>(gq16);
  auto rr = static_cast<const unsigned int*>(r);
  auto o16 = static_cast<unsigned int*>(gr16);
  const KeepSrc kp{reinterpret_cast<const unsigned short*>(keep), keep_seed, keep_threshold(keep_prob)};
  TOK_DISPATCH(cols / 128, {
    if (x_is_bf16)
      hipLaunchKernelGGL((add_ln_bwd<K, true>), grid, block, 0, st, g32, a16, aq, x, rr, weight, mean_rstd,
                         level_start, L, S, gx, o16, partials, rows, kp, keep_scale);
    else
      hipLaunchKernelGGL((add_ln_bwd<K, false>), grid, block, 0, st, g32, a16, aq, x, rr, weight, mean_rstd,
                         level_start, L, S, gx, o16, partials, rows, kp, keep_scale);
  });
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_relu_dropout_forward(const void* h, const unsigned char* keep, float keep_scale,
                                             const int* keep_seed, float keep_prob, void* y,
                                             long n, void* hip_stream) {
  if (!h || !y) return TRANSOAR_TOK_ERR_NULL;
  if (n <= 0 || (n & 7)) return TRANSOAR_TOK_ERR_DIM;
  const long n8 = n >> 3;
  hipLaunchKernelGGL(relu_dropout_fwd, dim3(static_cast<unsigned>((n8 + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(hip_stream), static_cast<const uint4*>(h),
                     KeepSrc{reinterpret_cast<const unsigned short*>(keep), keep_seed, keep_threshold(keep_prob)}, keep_scale,
                     static_cast<uint4*>(y), n8);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_relu_dropout_backward(const void* gy, const void* y, float keep_scale, void* gh, long n,
                                              void* hip_stream) {
  if (!gy || !y || !gh) return TRANSOAR_TOK_ERR_NULL;
  if (n <= 0 || (n & 7)) return TRANSOAR_TOK_ERR_DIM;
  const long n8 = n >> 3;
  hipLaunchKernelGGL(relu_dropout_bwd, dim3(static_cast<unsigned>((n8 + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(hip_stream), static_cast<const uint4*>(gy), static_cast<const uint4*>(y),
                     keep_scale, static_cast<uint4*>(gh), n8);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_sampling_head_forward(const void* proj, const float* ref, long ref_rows, const long* shapes,
                                             float* loc, float* attn, long tokens, int M, int L, int P,
                                             void* hip_stream) {
  if (!proj || !ref || !shapes || !loc || !attn) return TRANSOAR_TOK_ERR_NULL;
  if (tokens <= 0 || M <= 0 || L <= 0 || P <= 0 || L * P > 256 || ref_rows <= 0 || tokens % ref_rows) return TRANSOAR_TOK_ERR_DIM;
  const int G = L * P, threads = (256 / G) * G;
  const long n = tokens * M * G;
  hipLaunchKernelGGL(sampling_head_fwd, dim3(static_cast<unsigned>((n + threads - 1) / threads)), dim3(threads), 0,
                     static_cast<hipStream_t>(hip_stream), static_cast<const unsigned short*>(proj), ref, shapes, loc, attn,
                     M, L, P, n, ref_rows);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_sampling_head_backward(const float* g_loc, const float* g_attn, const float* attn,
                                              const long* shapes, void* g_proj, long tokens, int M, int L, int P,
                                              void* hip_stream) {
  if (!g_loc || !g_attn || !attn || !shapes || !g_proj) return TRANSOAR_TOK_ERR_NULL;
  if (tokens <= 0 || M <= 0 || L <= 0 || P <= 0 || L * P > 256) return TRANSOAR_TOK_ERR_DIM;
  const int G = L * P, threads = (256 / G) * G;
  const long n = tokens * M * G;
  hipLaunchKernelGGL(sampling_head_bwd, dim3(static_cast<unsigned>((n + threads - 1) / threads)), dim3(threads), 0,
                     static_cast<hipStream_t>(hip_stream), g_loc, g_attn, attn, shapes, static_cast<unsigned short*>(g_proj),
                     M, L, P, n);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_add_layernorm_partial_rows(void) { return kPersistentWaves; }
extern "C" int transoar_tokens_abi_version(void) { return 4; }
