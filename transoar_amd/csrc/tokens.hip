// Fused residual-add + LayerNorm (+ casts, + positional add) over the token
// stream of the refine block; C ABI in include/transoar_tokens.h.  gfx950 only.
//
// One wave per token row: cols = 128*K columns, lane l holds columns
// i*128 + 2l, +1 (i < K) -> every load/store of a row is one contiguous
// 512-byte (fp32) or 256-byte (bf16) wave access.  Statistics are two-pass in
// registers (mean, then centred second moment), fp32.  HBM-bound by design:
// forward moves 18 B/element (fp32 stream) where the unfused chain moves 48.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>

#include "../../include/transoar_tokens.h"

namespace {

constexpr int kWaves = 4;                       // rows per workgroup
constexpr int kPersistentWaves = 256 * 4 * kWaves;   // backward: 4 workgroups per CU (16 waves: the row walk is latency-bound)

__device__ __forceinline__ float bf16_lo(unsigned int u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned int f32_to_bf16_bits(float f) {      // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  return __builtin_bit_cast(unsigned short, static_cast<__bf16>(f));
}
__device__ __forceinline__ unsigned int pack_bf16(float a, float b) {      // (bf16(b) << 16) | bf16(a) in one instruction
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
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
  // one partial row per WORKGROUP (round 5: a row per wave = 4 096 rows of up to 2 304 floats, 37 MB for the column-sum launch
  // that follows): the waves add their sums into one LDS row in turn
  __shared__ float red[(2 + TRANSOAR_TOK_MAX_LEVELS) * cols];
  const int wv = threadIdx.x >> 6;
  for (int w = 0; w < kWaves; ++w) {
    if (wv == w) {
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const int c = i * 128 + 2 * lane;
        const float2 z{0.f, 0.f};
        const float2 a = w ? *reinterpret_cast<const float2*>(red + c) : z, b = w ? *reinterpret_cast<const float2*>(red + cols + c) : z;
        *reinterpret_cast<float2*>(red + c) = float2{a.x + dw[2 * i], a.y + dw[2 * i + 1]};
        *reinterpret_cast<float2*>(red + cols + c) = float2{b.x + db[2 * i], b.y + db[2 * i + 1]};
#pragma unroll
        for (int l = 0; l < TRANSOAR_TOK_MAX_LEVELS; ++l)
          if (l < L) {
            const float2 e = w ? *reinterpret_cast<const float2*>(red + (2 + l) * cols + c) : z;
            *reinterpret_cast<float2*>(red + (2 + l) * cols + c) = float2{e.x + dle[l][2 * i], e.y + dle[l][2 * i + 1]};
          }
      }
    }
    __syncthreads();
  }
  float* out = partials + static_cast<long>(blockIdx.x) * (2 + L) * cols;
  for (int c = threadIdx.x; c < (2 + L) * cols; c += 64 * kWaves) out[c] = red[c];
}

// Query of the FIRST refine layer: q16 = bf16(x + (pos_sine[s] + level_embed[level(s)])) from the bf16 pyramid tokens
// (the later layers get theirs from add_ln_fwd).  One thread per 8 consecutive columns: 16-byte loads of x, 2 x 16
// bytes of each fp32 operand, one 16-byte store.  Same rounding points as the eager chain (fp32 sum of the two
// positional terms, fp32 add, one rounding).
__global__ __launch_bounds__(256) void pos_query_fwd(const uint4* __restrict__ x16, const float* __restrict__ pos_sine,
                                                     const float* __restrict__ level_embed,
                                                     const int* __restrict__ level_start, int L, long S, int cols8,
                                                     uint4* __restrict__ q16, long n8) {
  const long g = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (g >= n8) return;
  const long row = g / cols8;
  const int c = static_cast<int>(g - row * cols8) * 8;
  const long srow = row % S;
  const int lvl = level_of(level_start, L, static_cast<int>(srow));
  const int cols = cols8 * 8;
  const uint4 x = x16[g];
  const float4 p0 = *reinterpret_cast<const float4*>(pos_sine + srow * cols + c);
  const float4 p1 = *reinterpret_cast<const float4*>(pos_sine + srow * cols + c + 4);
  const float4 e0 = *reinterpret_cast<const float4*>(level_embed + lvl * cols + c);
  const float4 e1 = *reinterpret_cast<const float4*>(level_embed + lvl * cols + c + 4);
  uint4 q;
  q.x = pack_bf16(bf16_lo(x.x) + (p0.x + e0.x), bf16_hi(x.x) + (p0.y + e0.y));
  q.y = pack_bf16(bf16_lo(x.y) + (p0.z + e0.z), bf16_hi(x.y) + (p0.w + e0.w));
  q.z = pack_bf16(bf16_lo(x.z) + (p1.x + e1.x), bf16_hi(x.z) + (p1.y + e1.y));
  q.w = pack_bf16(bf16_lo(x.w) + (p1.z + e1.z), bf16_hi(x.w) + (p1.w + e1.w));
  q16[g] = q;
}

// Its backward for level_embed: column sums of gq16 per level (the gradient of x is gq16 itself).  Persistent waves,
// one row per wave and trip (two rows in flight), the level sums in registers (the level is uniform per row), reduced
// over the workgroup's waves through LDS: partials (workgroups, L, cols) fp32, written completely.
template <int K>
__global__ __launch_bounds__(64 * kWaves) void pos_query_bwd(const unsigned int* __restrict__ gq16,
                                                             const int* __restrict__ level_start, int L, long S,
                                                             float* __restrict__ partials, long rows) {
  constexpr int cols = 128 * K;
  __shared__ float red[kWaves][cols];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long wave = static_cast<long>(blockIdx.x) * kWaves + wv;
  float dle[TRANSOAR_TOK_MAX_LEVELS][2 * K];
#pragma unroll
  for (int l = 0; l < TRANSOAR_TOK_MAX_LEVELS; ++l)
#pragma unroll
    for (int i = 0; i < 2 * K; ++i) dle[l][i] = 0.f;
  for (long row = wave; row < rows; row += 2 * kPersistentWaves) {
    const long row2 = row + kPersistentWaves;
    const bool two = row2 < rows;
    unsigned int u[K], v[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
      u[i] = gq16[(row * cols + i * 128 + 2 * lane) >> 1];
      v[i] = two ? gq16[(row2 * cols + i * 128 + 2 * lane) >> 1] : 0u;
    }
    const int lvl = level_of(level_start, L, static_cast<int>(row % S));
    const int lvl2 = two ? level_of(level_start, L, static_cast<int>(row2 % S)) : 0;
#pragma unroll
    for (int l = 0; l < TRANSOAR_TOK_MAX_LEVELS; ++l) {
      if (l == lvl) {
#pragma unroll
        for (int i = 0; i < K; ++i) { dle[l][2 * i] += bf16_lo(u[i]); dle[l][2 * i + 1] += bf16_hi(u[i]); }
      }
      if (l == lvl2) {
#pragma unroll
        for (int i = 0; i < K; ++i) { dle[l][2 * i] += bf16_lo(v[i]); dle[l][2 * i + 1] += bf16_hi(v[i]); }
      }
    }
  }
  float* out = partials + static_cast<long>(blockIdx.x) * L * cols;
#pragma unroll
  for (int l = 0; l < TRANSOAR_TOK_MAX_LEVELS; ++l) {
    if (l < L) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < K; ++i)
        *reinterpret_cast<float2*>(&red[wv][i * 128 + 2 * lane]) = float2{dle[l][2 * i], dle[l][2 * i + 1]};
      __syncthreads();
      for (int c = threadIdx.x; c < cols; c += 64 * kWaves) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) t += red[w][c];
        out[l * cols + c] = t;
      }
    }
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

// ---- the same two kernels for L = P = 4 (the flagship's 4 levels x 4 points), one thread per 16-BYTE PIECE of the
// projection row: a token's row is 6 M pieces of offsets (8 coordinates each -> 32 bytes of locations) and 2 M pieces of
// logits (half a head each; the two halves of a softmax sit in neighbouring lanes).  Consecutive threads read and write
// consecutive pieces, so every access of a wave is one contiguous run (the per-element kernels above moved 2 to 4 bytes
// per access behind three block barriers and two 64-bit divisions per element: 0.27 + 0.22 ms per call for 0.54 / 0.63 GB;
// one thread per (token, head) with 16-byte accesses 192 bytes apart was no faster).  Same arithmetic in the same order.
// A block holds a whole number of tokens; per-coordinate constants (bf16(size), index into the token's 12 reference
// coordinates) and the block's reference points are staged in LDS.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kHeadMaxTok = 32;                  // tokens per block (<= 256 / (8 M))

__global__ __launch_bounds__(256) void sampling_head_fwd_p16(const unsigned short* __restrict__ proj, const float* __restrict__ ref,
                                                             const long* __restrict__ shapes, float* __restrict__ loc,
                                                             float* __restrict__ attn, int M, unsigned tokens, unsigned ref_rows) {
  constexpr int L = 4, P = 4, G = 16;
  __shared__ float size_of[G * 3];               // by coordinate index inside a head: bf16((W, H, D)[k] of level l)
  __shared__ int ref_of[G * 3];                  // l * 3 + k
  __shared__ float ref_sh[kHeadMaxTok * L * 3];
  const int ppt = 8 * M, tpb = blockDim.x / ppt;             // pieces per token, tokens per block
  const unsigned t0 = blockIdx.x * tpb;
  if (threadIdx.x < G * 3) {
    const int idx = threadIdx.x, g = idx / 3, k = idx - 3 * g, l = g / P;
    size_of[idx] = bf16_round(static_cast<float>(shapes[l * 3 + 2 - k]));
    ref_of[idx] = l * 3 + k;
  }
  for (int i = threadIdx.x; i < tpb * L * 3; i += blockDim.x) {
    const unsigned t = t0 + i / (L * 3);
    ref_sh[i] = t < tokens ? ref[static_cast<long>(t % ref_rows) * (L * 3) + i % (L * 3)] : 0.f;
  }
  __syncthreads();
  const int tl = threadIdx.x / ppt, p = threadIdx.x - tl * ppt;
  const unsigned t = t0 + tl;
  if (tl >= tpb || t >= tokens) return;
  const u32x4_t v = reinterpret_cast<const u32x4_t*>(proj + static_cast<long>(t) * (4 * M * G))[p];
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
  float x[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) { x[2 * j] = bf16_lo(w[j]); x[2 * j + 1] = bf16_hi(w[j]); }
  if (p < 6 * M) {                               // offsets -> locations: ref + bf16(off / bf16(size))
    const int c = p % 6;                         // piece inside the head: coordinates 8 c .. 8 c + 7
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = ref_sh[tl * (L * 3) + ref_of[8 * c + j]] + bf16_round(x[j] / size_of[8 * c + j]);
    float4* dst = reinterpret_cast<float4*>(loc + static_cast<long>(t) * (M * G * 3) + p * 8);
    dst[0] = float4{r[0], r[1], r[2], r[3]};
    dst[1] = float4{r[4], r[5], r[6], r[7]};
  } else {                                       // logits -> softmax; the other half of the head is lane ^ 1
    const int q = p - 6 * M, half = q & 1;
    float mx = x[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) mx = fmaxf(mx, x[j]);
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = __expf(x[j] - mx); if (half == 0) s += x[j]; }
    const float s0 = __shfl_xor(s, 1);           // the first half's partial sum: the second half continues it in order
    if (half == 1) {
      s = s0;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += x[j];
    }
    const float other = __shfl_xor(s, 1);
    const float sum = half == 1 ? s : other;
    float4* dst = reinterpret_cast<float4*>(attn + static_cast<long>(t) * (M * G) + q * 8);
    dst[0] = float4{x[0] / sum, x[1] / sum, x[2] / sum, x[3] / sum};
    dst[1] = float4{x[4] / sum, x[5] / sum, x[6] / sum, x[7] / sum};
  }
}

__global__ __launch_bounds__(256) void sampling_head_bwd_p16(const float* __restrict__ g_loc, const float* __restrict__ g_attn,
                                                             const float* __restrict__ attn, const long* __restrict__ shapes,
                                                             unsigned short* __restrict__ g_proj, int M, unsigned tokens) {
  constexpr int P = 4, G = 16;
  __shared__ float size_of[G * 3];
  const int ppt = 8 * M, tpb = blockDim.x / ppt;
  if (threadIdx.x < G * 3) {
    const int idx = threadIdx.x, g = idx / 3, k = idx - 3 * g, l = g / P;
    size_of[idx] = bf16_round(static_cast<float>(shapes[l * 3 + 2 - k]));
  }
  __syncthreads();
  const int tl = threadIdx.x / ppt, p = threadIdx.x - tl * ppt;
  const unsigned t = blockIdx.x * tpb + tl;
  if (tl >= tpb || t >= tokens) return;
  unsigned h[8];
  if (p < 6 * M) {
    const int c = p % 6;
    const float4* src = reinterpret_cast<const float4*>(g_loc + static_cast<long>(t) * (M * G * 3) + p * 8);
    const float4 u = src[0], v = src[1];
    const float gl[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = f32_to_bf16_bits(bf16_round(gl[j]) / size_of[8 * c + j]);
  } else {
    const int q = p - 6 * M, half = q & 1;
    const long e = static_cast<long>(t) * (M * G) + q * 8;
    const float4 a0 = *reinterpret_cast<const float4*>(attn + e), a1 = *reinterpret_cast<const float4*>(attn + e + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(g_attn + e), b1 = *reinterpret_cast<const float4*>(g_attn + e + 4);
    const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float ga[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float d = 0.f;
    if (half == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) d += a[j] * ga[j];
    }
    const float d0 = __shfl_xor(d, 1);
    if (half == 1) {
      d = d0;
#pragma unroll
      for (int j = 0; j < 8; ++j) d += a[j] * ga[j];
    }
    const float other = __shfl_xor(d, 1);
    const float dot = half == 1 ? d : other;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = f32_to_bf16_bits(a[j] * (ga[j] - dot));
  }
  reinterpret_cast<u32x4_t*>(g_proj + static_cast<long>(t) * (4 * M * G))[p] =
      u32x4_t{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
}

}  // namespace


// ---------------------------------------------------------------------------
// LayerNorm over SHORT rows (48 .. 512 channels, a multiple of 8; up to 1536 on the *_wide kernels below): the norm1 / norm2 / patch-merge norms of the Swin
// encoder stages (transoar/models/backbones/encoder_blocks.py:143-327, nn.LayerNorm over 48 / 96 / 192 / 384 channels
// of 10^5 .. 10^6 tokens).  aten's kernels take 0.4 ms per pass on the 1.6 M x 48 fp32 rows of stage 2 (23 ms per step);
// add_ln_* above needs a multiple of 128 columns.  Here a GROUP of G lanes (the next power of two >= cols / 8) owns a row,
// 8 channels per lane in registers, statistics by xor-shuffles inside the group; output bf16 (what the following
// projection rounds to under autocast).  Backward: dx in the input's type, and per-lane partial sums of the weight /
// bias gradients over a grid-stride loop of rows -> partials (n_partial_rows, 2 * cols) for one column-sum launch.
// ---------------------------------------------------------------------------
constexpr int kLnBlocks = 1024;          // persistent workgroups of the backward (4 waves each)

template <bool XBF>
__device__ __forceinline__ void ln_load8(const void* x, long row, int cols, int c0, float (&v)[8]) {
  if (XBF) {
    const uint4 q = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(x) + row * cols + c0);
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = bf16_lo(w[i]); v[2 * i + 1] = bf16_hi(w[i]); }
  } else {
    const float4* pf = reinterpret_cast<const float4*>(static_cast<const float*>(x) + row * cols + c0);
    const float4 a = pf[0], b = pf[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int d = 1; d < G; d <<= 1) v += __shfl_xor(v, d, 64);
  return v;
}

template <int G, bool XBF>
__global__ __launch_bounds__(256) void ln_rows_fwd(const void* __restrict__ x, const float* __restrict__ weight, const float* __restrict__ bias,
                                                   float eps, uint4* __restrict__ y16, float* __restrict__ mean_out,
                                                   float* __restrict__ rstd_out, long rows, int cols) {
  constexpr int RPW = 64 / G;                                     // rows per wave
  const int lane = threadIdx.x & 63;
  const int sub = lane / G, gl = lane % G;
  const long row = (static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * RPW + sub;
  const int c0 = gl * 8;
  const bool on = row < rows && c0 < cols;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (on) ln_load8<XBF>(x, row, cols, c0, v);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  const float mean = group_sum<G>(s) / cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { const float d = on ? v[i] - mean : 0.f; q += d * d; }
  const float rstd = rsqrtf(group_sum<G>(q) / cols + eps);
  if (!on) return;
  const float4 w0 = *reinterpret_cast<const float4*>(weight + c0), w1 = *reinterpret_cast<const float4*>(weight + c0 + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(bias + c0), b1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
  const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  float o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (v[i] - mean) * rstd * ww[i] + bb[i];
  y16[(row * cols + c0) >> 3] = uint4{pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
  if (gl == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

template <int G, bool XBF>
__global__ __launch_bounds__(256) void ln_rows_bwd(const uint4* __restrict__ g16, const void* __restrict__ x, const float* __restrict__ weight,
                                                   const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                   const void* __restrict__ dx_add, void* __restrict__ dx,
                                                   float* __restrict__ partials, long rows, int cols) {
  constexpr int RPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int sub = lane / G, gl = lane % G;
  const int c0 = gl * 8;
  const bool col_on = c0 < cols;
  float ww[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col_on) {
    const float4 w0 = *reinterpret_cast<const float4*>(weight + c0), w1 = *reinterpret_cast<const float4*>(weight + c0 + 4);
    ww[0] = w0.x; ww[1] = w0.y; ww[2] = w0.z; ww[3] = w0.w; ww[4] = w1.x; ww[5] = w1.y; ww[6] = w1.z; ww[7] = w1.w;
  }
  float dw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, db[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const long wave_id = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const long n_waves = static_cast<long>(gridDim.x) * 4;
  for (long r0 = wave_id * RPW; r0 < rows; r0 += n_waves * RPW) {
    const long row = r0 + sub;
    const bool on = row < rows && col_on;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float mean = 0.f, rstd = 0.f;
    if (on) {
      ln_load8<XBF>(x, row, cols, c0, v);
      const uint4 q = g16[(row * cols + c0) >> 3];
      const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { g[2 * i] = bf16_lo(w4[i]); g[2 * i + 1] = bf16_hi(w4[i]); }
      mean = mean_in[row];
      rstd = rstd_in[row];
    }
    float xh[8], gw[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      xh[i] = (v[i] - mean) * rstd;
      gw[i] = g[i] * ww[i];
      s1 += gw[i];
      s2 += gw[i] * xh[i];
      dw[i] += g[i] * xh[i];
      db[i] += g[i];
    }
    const float c1 = group_sum<G>(s1) / cols, c2 = group_sum<G>(s2) / cols;
    if (on) {
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = rstd * (gw[i] - c1 - xh[i] * c2);
      if (dx_add != nullptr) {                       // the gradient that reaches x past the norm (a residual branch): summed here
        float a[8];
        ln_load8<XBF>(dx_add, row, cols, c0, a);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] += a[i];
      }
      if (XBF) {
        reinterpret_cast<uint4*>(dx)[(row * cols + c0) >> 3] =
            uint4{pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
      } else {
        float4* po = reinterpret_cast<float4*>(static_cast<float*>(dx) + row * cols + c0);
        po[0] = float4{o[0], o[1], o[2], o[3]};
        po[1] = float4{o[4], o[5], o[6], o[7]};
      }
    }
  }
  // the RPW row groups of a wave hold partial sums of the same columns: add them, then one partial row per wave
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int d = G; d < 64; d <<= 1) { dw[i] += __shfl_xor(dw[i], d, 64); db[i] += __shfl_xor(db[i], d, 64); }
  }
  // ... and the four waves of the workgroup meet in LDS: ONE partial row per workgroup (a row per wave made the column-sum
  // launch that follows walk 4 096 rows through 24 .. 96 workgroups: 16 us each, 28 of them per Swin step)
  __shared__ float wg_red[4][2 * 512];
  const int wv = threadIdx.x >> 6;
  if (sub == 0 && col_on) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { wg_red[wv][c0 + i] = dw[i]; wg_red[wv][cols + c0 + i] = db[i]; }
  }
  __syncthreads();
  float* pr = partials + static_cast<long>(blockIdx.x) * (2L * cols);
  for (int c = threadIdx.x; c < 2 * cols; c += 256) pr[c] = (wg_red[0][c] + wg_red[1][c]) + (wg_red[2][c] + wg_red[3][c]);
}

// The same two kernels for 512 < cols <= 1536 (the patch-merge norms of the later Swin stages: 8 x 96 = 768 and 8 x 192 = 1536
// channels): one row per wave, up to three 8-channel pieces per lane (columns 8 lane + 512 ch).
constexpr int kLnWideCh = 3;
template <bool XBF>
__global__ __launch_bounds__(256) void ln_rows_fwd_wide(const void* __restrict__ x, const float* __restrict__ weight, const float* __restrict__ bias,
                                                        float eps, uint4* __restrict__ y16, float* __restrict__ mean_out,
                                                        float* __restrict__ rstd_out, long rows, int cols) {
  const int lane = threadIdx.x & 63;
  const long row = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[kLnWideCh][8];
  float s = 0.f;
#pragma unroll
  for (int ch = 0; ch < kLnWideCh; ++ch) {
    const int c0 = lane * 8 + 512 * ch;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[ch][i] = 0.f;
    if (c0 < cols) ln_load8<XBF>(x, row, cols, c0, v[ch]);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[ch][i];
  }
  const float mean = group_sum<64>(s) / cols;
  float q = 0.f;
#pragma unroll
  for (int ch = 0; ch < kLnWideCh; ++ch)
    if (lane * 8 + 512 * ch < cols) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = v[ch][i] - mean; q += d * d; }
    }
  const float rstd = rsqrtf(group_sum<64>(q) / cols + eps);
#pragma unroll
  for (int ch = 0; ch < kLnWideCh; ++ch) {
    const int c0 = lane * 8 + 512 * ch;
    if (c0 >= cols) continue;
    const float4 w0 = *reinterpret_cast<const float4*>(weight + c0), w1 = *reinterpret_cast<const float4*>(weight + c0 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(bias + c0), b1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
    const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (v[ch][i] - mean) * rstd * ww[i] + bb[i];
    y16[(row * cols + c0) >> 3] = uint4{pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
  }
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

template <bool XBF>
__global__ __launch_bounds__(256) void ln_rows_bwd_wide(const uint4* __restrict__ g16, const void* __restrict__ x, const float* __restrict__ weight,
                                                        const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                        const void* __restrict__ dx_add, void* __restrict__ dx,
                                                        float* __restrict__ partials, long rows, int cols) {
  const int lane = threadIdx.x & 63;
  float ww[kLnWideCh][8], dw[kLnWideCh][8], db[kLnWideCh][8];
#pragma unroll
  for (int ch = 0; ch < kLnWideCh; ++ch) {
    const int c0 = lane * 8 + 512 * ch;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ww[ch][i] = 0.f; dw[ch][i] = 0.f; db[ch][i] = 0.f; }
    if (c0 < cols) {
      const float4 w0 = *reinterpret_cast<const float4*>(weight + c0), w1 = *reinterpret_cast<const float4*>(weight + c0 + 4);
      ww[ch][0] = w0.x; ww[ch][1] = w0.y; ww[ch][2] = w0.z; ww[ch][3] = w0.w; ww[ch][4] = w1.x; ww[ch][5] = w1.y; ww[ch][6] = w1.z; ww[ch][7] = w1.w;
    }
  }
  const long wave_id = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const long n_waves = static_cast<long>(gridDim.x) * 4;
  for (long row = wave_id; row < rows; row += n_waves) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    float xh[kLnWideCh][8], gw[kLnWideCh][8], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ch = 0; ch < kLnWideCh; ++ch) {
      const int c0 = lane * 8 + 512 * ch;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const bool on = c0 < cols;
      if (on) {
        ln_load8<XBF>(x, row, cols, c0, v);
        const uint4 q = g16[(row * cols + c0) >> 3];
        const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { g[2 * i] = bf16_lo(w4[i]); g[2 * i + 1] = bf16_hi(w4[i]); }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        xh[ch][i] = on ? (v[i] - mean) * rstd : 0.f;
        gw[ch][i] = g[i] * ww[ch][i];
        s1 += gw[ch][i];
        s2 += gw[ch][i] * xh[ch][i];
        dw[ch][i] += g[i] * xh[ch][i];
        db[ch][i] += g[i];
      }
    }
    const float c1 = group_sum<64>(s1) / cols, c2 = group_sum<64>(s2) / cols;
#pragma unroll
    for (int ch = 0; ch < kLnWideCh; ++ch) {
      const int c0 = lane * 8 + 512 * ch;
      if (c0 >= cols) continue;
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = rstd * (gw[ch][i] - c1 - xh[ch][i] * c2);
      if (dx_add != nullptr) {
        float a[8];
        ln_load8<XBF>(dx_add, row, cols, c0, a);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] += a[i];
      }
      if (XBF) {
        reinterpret_cast<uint4*>(dx)[(row * cols + c0) >> 3] =
            uint4{pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
      } else {
        float4* po = reinterpret_cast<float4*>(static_cast<float*>(dx) + row * cols + c0);
        po[0] = float4{o[0], o[1], o[2], o[3]};
        po[1] = float4{o[4], o[5], o[6], o[7]};
      }
    }
  }
  // one partial row per workgroup (see ln_rows_bwd): the four waves meet in LDS
  __shared__ float wg_red[4][2 * 512 * kLnWideCh];
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int ch = 0; ch < kLnWideCh; ++ch) {
    const int c0 = lane * 8 + 512 * ch;
    if (c0 >= cols) continue;
#pragma unroll
    for (int i = 0; i < 8; ++i) { wg_red[wv][c0 + i] = dw[ch][i]; wg_red[wv][cols + c0 + i] = db[ch][i]; }
  }
  __syncthreads();
  float* pr = partials + static_cast<long>(blockIdx.x) * (2L * cols);
  for (int c = threadIdx.x; c < 2 * cols; c += 256) pr[c] = (wg_red[0][c] + wg_red[1][c]) + (wg_red[2][c] + wg_red[3][c]);
}

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
  if (L == 4 && P == 4 && 8 * M <= 256 && tokens < (1L << 31) && ref_rows < (1L << 31)) {
    const int tpb = 256 / (8 * M), threads_p = tpb * 8 * M;     // a whole number of tokens per block
    hipLaunchKernelGGL(sampling_head_fwd_p16, dim3(static_cast<unsigned>((tokens + tpb - 1) / tpb)), dim3(threads_p), 0,
                       static_cast<hipStream_t>(hip_stream), static_cast<const unsigned short*>(proj), ref, shapes, loc, attn, M,
                       static_cast<unsigned>(tokens), static_cast<unsigned>(ref_rows));
    return static_cast<int>(hipGetLastError());
  }
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
  if (L == 4 && P == 4 && 8 * M <= 256 && tokens < (1L << 31)) {
    const int tpb = 256 / (8 * M), threads_p = tpb * 8 * M;
    hipLaunchKernelGGL(sampling_head_bwd_p16, dim3(static_cast<unsigned>((tokens + tpb - 1) / tpb)), dim3(threads_p), 0,
                       static_cast<hipStream_t>(hip_stream), g_loc, g_attn, attn, shapes, static_cast<unsigned short*>(g_proj), M,
                       static_cast<unsigned>(tokens));
    return static_cast<int>(hipGetLastError());
  }
  hipLaunchKernelGGL(sampling_head_bwd, dim3(static_cast<unsigned>((n + threads - 1) / threads)), dim3(threads), 0,
                     static_cast<hipStream_t>(hip_stream), g_loc, g_attn, attn, shapes, static_cast<unsigned short*>(g_proj),
                     M, L, P, n);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_pos_query_forward(const void* x16, const float* pos_sine, const float* level_embed,
                                          const int* level_start, int L, long S, void* q16, long rows, int cols,
                                          void* hip_stream) {
  if (!x16 || !pos_sine || !level_embed || !level_start || !q16) return TRANSOAR_TOK_ERR_NULL;
  if (rows <= 0 || cols <= 0 || cols % 128 || cols > 1024 || S <= 0) return TRANSOAR_TOK_ERR_DIM;
  if (L <= 0 || L > TRANSOAR_TOK_MAX_LEVELS) return TRANSOAR_TOK_ERR_LEVELS;
  const long n8 = rows * (cols / 8);
  hipLaunchKernelGGL(pos_query_fwd, dim3(static_cast<unsigned>((n8 + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(hip_stream), static_cast<const uint4*>(x16), pos_sine, level_embed,
                     level_start, L, S, cols / 8, static_cast<uint4*>(q16), n8);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_pos_query_backward(const void* gq16, const int* level_start, int L, long S, float* partials,
                                           long rows, int cols, void* hip_stream) {
  if (!gq16 || !level_start || !partials) return TRANSOAR_TOK_ERR_NULL;
  if (rows <= 0 || cols <= 0 || cols % 128 || cols > 1024 || S <= 0) return TRANSOAR_TOK_ERR_DIM;
  if (L <= 0 || L > TRANSOAR_TOK_MAX_LEVELS) return TRANSOAR_TOK_ERR_LEVELS;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const dim3 grid(kPersistentWaves / kWaves), block(64 * kWaves);
  auto g = static_cast<const unsigned int*>(gq16);
  TOK_DISPATCH(cols / 128, {
    hipLaunchKernelGGL((pos_query_bwd<K>), grid, block, 0, st, g, level_start, L, S, partials, rows);
  });
  return static_cast<int>(hipGetLastError());
}


static int ln_group(int cols) {
  const int lanes = cols / 8;
  int g = 1;
  while (g < lanes) g <<= 1;
  return g;
}
#define LN_DISPATCH(G_, BODY)                       \
  switch (G_) {                                     \
    case 8: { constexpr int G = 8; BODY; } break;   \
    case 16: { constexpr int G = 16; BODY; } break; \
    case 32: { constexpr int G = 32; BODY; } break; \
    default: { constexpr int G = 64; BODY; } break; \
  }

extern "C" int transoar_ln_rows_forward(const void* x, int x_is_bf16, const float* weight, const float* bias, float eps, void* y16,
                                        float* mean, float* rstd, long rows, int cols, void* hip_stream) {
  if (!x || !weight || !bias || !y16 || !mean || !rstd) return TRANSOAR_TOK_ERR_NULL;
  if (rows <= 0 || cols < 8 || cols > 512 * kLnWideCh || (cols & 7)) return TRANSOAR_TOK_ERR_DIM;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  if (cols > 512) {
    const dim3 wgrid(static_cast<unsigned>((rows + 3) / 4));
    if (x_is_bf16) hipLaunchKernelGGL((ln_rows_fwd_wide<true>), wgrid, dim3(256), 0, st, x, weight, bias, eps, static_cast<uint4*>(y16), mean, rstd, rows, cols);
    else hipLaunchKernelGGL((ln_rows_fwd_wide<false>), wgrid, dim3(256), 0, st, x, weight, bias, eps, static_cast<uint4*>(y16), mean, rstd, rows, cols);
    return static_cast<int>(hipGetLastError());
  }
  const int g = std::max(8, ln_group(cols));
  const long rows_per_block = 4L * (64 / g);
  const dim3 grid(static_cast<unsigned>((rows + rows_per_block - 1) / rows_per_block));
  LN_DISPATCH(g, {
    if (x_is_bf16) hipLaunchKernelGGL((ln_rows_fwd<G, true>), grid, dim3(256), 0, st, x, weight, bias, eps, static_cast<uint4*>(y16), mean, rstd, rows, cols);
    else hipLaunchKernelGGL((ln_rows_fwd<G, false>), grid, dim3(256), 0, st, x, weight, bias, eps, static_cast<uint4*>(y16), mean, rstd, rows, cols);
  });
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_ln_rows_partial_rows(void) { return kLnBlocks; }

extern "C" int transoar_ln_rows_backward(const void* g16, const void* x, int x_is_bf16, const float* weight, const float* mean,
                                         const float* rstd, const void* dx_add, void* dx, float* partials, long rows, int cols,
                                         void* hip_stream) {
  if (!g16 || !x || !weight || !mean || !rstd || !dx || !partials) return TRANSOAR_TOK_ERR_NULL;
  if (rows <= 0 || cols < 8 || cols > 512 * kLnWideCh || (cols & 7)) return TRANSOAR_TOK_ERR_DIM;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  if (cols > 512) {
    if (x_is_bf16) hipLaunchKernelGGL((ln_rows_bwd_wide<true>), dim3(kLnBlocks), dim3(256), 0, st, static_cast<const uint4*>(g16), x, weight, mean, rstd, dx_add, dx, partials, rows, cols);
    else hipLaunchKernelGGL((ln_rows_bwd_wide<false>), dim3(kLnBlocks), dim3(256), 0, st, static_cast<const uint4*>(g16), x, weight, mean, rstd, dx_add, dx, partials, rows, cols);
    return static_cast<int>(hipGetLastError());
  }
  const int g = std::max(8, ln_group(cols));
  LN_DISPATCH(g, {
    if (x_is_bf16) hipLaunchKernelGGL((ln_rows_bwd<G, true>), dim3(kLnBlocks), dim3(256), 0, st, static_cast<const uint4*>(g16), x, weight, mean, rstd, dx_add, dx, partials, rows, cols);
    else hipLaunchKernelGGL((ln_rows_bwd<G, false>), dim3(kLnBlocks), dim3(256), 0, st, static_cast<const uint4*>(g16), x, weight, mean, rstd, dx_add, dx, partials, rows, cols);
  });
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_pos_query_partial_rows(void) { return kPersistentWaves / kWaves; }

extern "C" int transoar_add_layernorm_partial_rows(void) { return kPersistentWaves / kWaves; }
extern "C" int transoar_tokens_abi_version(void) { return 7; }
