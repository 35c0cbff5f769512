// Shared device helpers for the gfx950 MSDeformAttn-3D kernels.
// wave = 64 lanes everywhere in this directory (CDNA4); no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace transoar {

constexpr int kWavesPerBlock = 4;   // 256-thread workgroups
constexpr int kChunk = 16;          // sampling points per location-load round

struct bf16_t { unsigned short bits; };
struct f16_t { _Float16 v; };

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

__device__ __forceinline__ float bf16_to_f32(unsigned short b) {
  return __uint_as_float(static_cast<unsigned int>(b) << 16);
}
// round-to-nearest-even, NaN kept quiet (matches torch's float->bfloat16)
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<unsigned short>((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<unsigned short>(u >> 16);
}

// Storage-type traits: accumulator type, elements per 16-byte lane vector,
// scalar load/store and 16-byte vector load/store with conversion.
template <typename T> struct Elem;

template <> struct Elem<float> {
  using acc = float;
  static constexpr int VEC = 4;
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ void unpack(const u32x4& r, float (&o)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = __uint_as_float(r[i]);
  }
  static __device__ __forceinline__ u32x4 pack(const float (&o)[4]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __float_as_uint(o[i]);
    return r;
  }
};

template <> struct Elem<double> {
  using acc = double;
  static constexpr int VEC = 2;
  static __device__ __forceinline__ double ld(const double* p) { return *p; }
  static __device__ __forceinline__ void st(double* p, double v) { *p = v; }
  static __device__ __forceinline__ void unpack(const u32x4& r, double (&o)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      o[i] = __longlong_as_double((static_cast<long long>(r[2 * i + 1]) << 32) |
                                  static_cast<long long>(r[2 * i]));
  }
  static __device__ __forceinline__ u32x4 pack(const double (&o)[2]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long b = __double_as_longlong(o[i]);
      r[2 * i] = static_cast<unsigned int>(b);
      r[2 * i + 1] = static_cast<unsigned int>(b >> 32);
    }
    return r;
  }
};

template <> struct Elem<bf16_t> {
  using acc = float;
  static constexpr int VEC = 8;
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(p->bits); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { p->bits = f32_to_bf16(v); }
  static __device__ __forceinline__ void unpack(const u32x4& r, float (&o)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(r[i] << 16);
      o[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ u32x4 pack(const float (&o)[8]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      r[i] = static_cast<unsigned int>(f32_to_bf16(o[2 * i])) |
             (static_cast<unsigned int>(f32_to_bf16(o[2 * i + 1])) << 16);
    return r;
  }
  // acc + a.lo*b.lo + a.hi*b.hi on two packed pairs, fp32 accumulate (v_dot2c_f32_bf16)
  static __device__ __forceinline__ float dot2(unsigned int a, unsigned int b, float acc) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), acc, false);
  }
};

template <> struct Elem<f16_t> {
  using acc = float;
  static constexpr int VEC = 8;
  static __device__ __forceinline__ float h2f(unsigned int bits16) {
    return static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>(bits16)));
  }
  static __device__ __forceinline__ unsigned int f2h(float f) {
    return __builtin_bit_cast(unsigned short, static_cast<_Float16>(f));
  }
  static __device__ __forceinline__ float ld(const f16_t* p) { return static_cast<float>(p->v); }
  static __device__ __forceinline__ void st(f16_t* p, float v) { p->v = static_cast<_Float16>(v); }
  static __device__ __forceinline__ void unpack(const u32x4& r, float (&o)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = h2f(r[i] & 0xffffu);
      o[2 * i + 1] = h2f(r[i] >> 16);
    }
  }
  static __device__ __forceinline__ u32x4 pack(const float (&o)[8]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = f2h(o[2 * i]) | (f2h(o[2 * i + 1]) << 16);
    return r;
  }
  static __device__ __forceinline__ float dot2(unsigned int a, unsigned int b, float acc) {   // v_dot2_f32_f16
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), acc, false);
  }
};

// pixel coordinate of a normalised location: loc*size - 0.5 with the multiply
// and the subtract rounded separately (no FMA contraction), so floor() picks
// the same cell as the scalar oracle (oracle/msda3d_oracle_impl.h) bit for bit.
// (HIP's __fmul_rn is a plain '*' and would still be contracted.)
__device__ __forceinline__ float pixel_coord(float loc, int size) {
#pragma clang fp contract(off)
  const float t = loc * static_cast<float>(size);
  return t - 0.5f;
}
__device__ __forceinline__ double pixel_coord(double loc, int size) {
#pragma clang fp contract(off)
  const double t = loc * static_cast<double>(size);
  return t - 0.5;
}

// wave-uniform broadcast of lane `src`'s value (src must be wave-uniform)
__device__ __forceinline__ float bcast(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ double bcast(double v, int src) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane(static_cast<int>(b), src);
  const unsigned hi = __builtin_amdgcn_readlane(static_cast<int>(b >> 32), src);
  return __longlong_as_double((static_cast<long long>(hi) << 32) | lo);
}

__device__ __forceinline__ float xor_lanes(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ double xor_lanes(double v, int mask) { return __shfl_xor(v, mask, 64); }

// value of lane `src` (per-lane index, ds_bpermute)
__device__ __forceinline__ float xor_free_shfl(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ double xor_free_shfl(double v, int src) { return __shfl(v, src, 64); }

// hardware fp atomics (no CAS loop): compiled with -munsafe-fp-atomics
__device__ __forceinline__ void atomic_accum(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void atomic_accum(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sum-transpose of 64 per-lane partials over the 64 lanes of a wave: on return
// lane r holds  sum over lanes of v[r].  63 exchange+add steps (vs 64*6 for 64
// separate butterflies).  Each fold writes a fresh array: folding in place
// keeps hipcc (ROCm 7.2) from promoting the 64-float array to registers
// (272 B/lane of scratch instead).
template <int HALF, typename A>
__device__ __forceinline__ void fold_lanes(const A (&in)[2 * HALF], A (&out)[HALF], int lane) {
  const bool up = (lane & HALF) != 0;
#pragma unroll
  for (int i = 0; i < HALF; ++i) {
    const A send = up ? in[i] : in[i + HALF];
    const A keep = up ? in[i + HALF] : in[i];
    out[i] = keep + xor_lanes(send, HALF);
  }
}
template <typename A>
__device__ __forceinline__ A sum_transpose64(const A (&v)[64], int lane) {
  A a32[32], a16[16], a8[8], a4[4], a2[2], a1[1];
  fold_lanes<32>(v, a32, lane);
  fold_lanes<16>(a32, a16, lane);
  fold_lanes<8>(a16, a8, lane);
  fold_lanes<4>(a8, a4, lane);
  fold_lanes<2>(a4, a2, lane);
  fold_lanes<1>(a2, a1, lane);
  return a1[0];
}

// Work ordering.  Waves that run close together in time should touch
// neighbouring voxels in all three axes, so that the 8-corner / 8-cell
// footprints they share are still in L2 (4 MiB per XCD): linear (d,h,w) order
// only gives reuse along w and h; a d-neighbour is 15 000 waves away.  When the
// host knows the level shapes, work is therefore walked in 4x4x8-voxel bricks
// (128 slots per brick, padded at the level borders).  Purely a schedule:
// results do not depend on it.
constexpr int kBrickD = 4, kBrickH = 4, kBrickW = 8, kBrickSlots = 128;
struct BrickOrder {
  int enabled, L;
  int D[8], H[8], W[8], start[8];
  int nbh[8], nbw[8];
  int pad_start[9];       // first padded slot of each level, pad_start[L] = total
};
// padded slot -> voxel row of the pyramid, or -1 for a padding slot (x wave-uniform)
__device__ __forceinline__ int brick_slot_to_row(const BrickOrder& o, int x) {
  int l = 0;
  for (int t = 1; t < o.L; ++t) l += (x >= o.pad_start[t]) ? 1 : 0;
  const int local = x - o.pad_start[l];
  const int brick = local >> 7, slot = local & 127;
  const int bw = brick % o.nbw[l];
  const int r = brick / o.nbw[l];
  const int bh = r % o.nbh[l], bd = r / o.nbh[l];
  const int d = bd * kBrickD + (slot >> 5), h = bh * kBrickH + ((slot >> 3) & 3), w = bw * kBrickW + (slot & 7);
  if (d >= o.D[l] || h >= o.H[l] || w >= o.W[l]) return -1;
  return o.start[l] + (d * o.H[l] + h) * o.W[l] + w;
}
// work unit u of a (batch, row-or-query, head) decomposition -> flat (b*R + r)*M + m, or -1.
// R = rows per batch element (S or Lq).
__device__ __forceinline__ long ordered_unit(const BrickOrder& o, long u, int R, int M) {
  if (!o.enabled) return u;
  const int total = o.pad_start[o.L];
  int m, x;
  long b;
  if (o.enabled == 2) {        // brick-major: all 128 slots of a brick for one head, then the next head
    const int slot = static_cast<int>(u & 127);
    const long t = u >> 7;
    m = static_cast<int>(t % M);
    const long t2 = t / M;
    const int bricks = total >> 7;
    x = static_cast<int>(t2 % bricks) * 128 + slot;
    b = t2 / bricks;
  } else {                     // head-minor: the heads of one voxel are consecutive waves
    m = static_cast<int>(u % M);
    const long t = u / M;
    x = static_cast<int>(t % total);
    b = t / total;
  }
  const int r = brick_slot_to_row(o, x);
  return r < 0 ? -1 : (b * R + r) * M + m;
}

// Observed (not contractual) placement: block b runs on XCD b % 8.  Remap so
// each XCD walks one contiguous eighth of the work and neighbouring query
// blocks share that XCD's L2.  Returns -1 for the padding blocks.
__device__ __forceinline__ long xcd_contiguous_block(long bid, long nblk) {
  const long per = (nblk + 7) >> 3;
  const long swz = (bid & 7) * per + (bid >> 3);
  return swz < nblk ? swz : -1;
}

// Record of a sorted point for the matrix-core walks: 16 bytes, ONE write per point.  Round 2's record was 32 bytes
// (four products a * f_d * f_h, l_w, the row index, 8 bytes of padding): 0.72 GB written and read back per call at the
// flagship size.  The three fractions l_d, l_h, l_w in [0, 1) are stored as 21-bit fixed point (2^-22 absolute error:
// finer than the fp32 rounding of the pixel coordinate they are cut from once a level has more than 32 voxels on an
// axis), the attention weight and the grad_out row stay 32 bits.  The 8 corner weights are rebuilt when a record is
// staged: t[2 dd + dh] = a * f_d * f_h, then t * (1 - l_w) and t * l_w.
struct alignas(16) PointR16 {
  float a;
  int item;          // row of grad_out, (b * Lq + q) * M + m
  unsigned lo, hi;   // l_d | l_h << 21 | l_w << 42
};
constexpr float kFracScale = 2097152.f;          // 2^21
__device__ __forceinline__ unsigned frac21(float f) {
  return min(static_cast<unsigned>(f * kFracScale + 0.5f), 0x1fffffu);
}
__device__ __forceinline__ PointR16 make_point_r16(float a, int item, float ld, float lh, float lw) {
  const unsigned qd = frac21(ld), qh = frac21(lh), qw = frac21(lw);
  return PointR16{a, item, qd | (qh << 21), (qh >> 11) | (qw << 10)};
}
__device__ __forceinline__ void point_r16_fracs(unsigned lo, unsigned hi, float& ld, float& lh, float& lw) {
  constexpr float inv = 1.f / kFracScale;
  ld = static_cast<float>(lo & 0x1fffffu) * inv;
  lh = static_cast<float>((lo >> 21) | ((hi & 0x3ffu) << 11)) * inv;
  lw = static_cast<float>(hi >> 10) * inv;
}

}  // namespace transoar
