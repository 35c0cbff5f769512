// Forward gather of MSDeformAttn-3D on the matrix cores, point-column form (gfx950).
//
// Semantics: SURVEY.md appendix A (ops/src/cuda/ms_deform_im2col_cuda.cuh:31-114, 370-439); with FUSED the
// sampling head of the module as well (ops/modules/ms_deform_attn.py:114-128: softmax over the L*P logits and
// ref + offset / (W, H, D)), read straight from the stacked bf16 projection -- SURVEY 8(d) counts the bytes
// of the logits/offsets actually read in that case.
//
// msda3d_mma.hpp (round 2) gave a wave 32 queries and made the 32 MFMA columns the QUERIES: the four points
// of a query then share a column, their corners may coincide, and the weight block had to be an fp32
// read-add-write structure in LDS, rebuilt K-block by K-block and split into bf16 hi + lo terms on its way
// into every MFMA (4 235 VALU instructions per wave, two thirds of them bookkeeping).  Here
//
//   * one WAVE owns 8 queries (a 2x2x2 sub-brick of the pyramid) of one head, and the 32 MFMA columns are
//     its 8 x 4 (query, point) pairs: a column holds ONE point per level, whose 8 corners are 8 distinct
//     rows -- the weight block is written, never accumulated: each entry is split once into bf16 hi | lo
//     halves of one 32-bit word when it is made, and stored with one ds_write_b32;
//   * lane = (kh, q, p): the lane computes the geometry of point p of query q for the two levels 2kh, 2kh+1
//     (sampling head, pixel coordinates, trilinear weights), the 32 lanes of a half reduce the two levels'
//     corner boxes together (packed 16-bit DPP minima), and the two halves then trade half of their entries
//     (v_permlane32_swap) so that lane (kh, q, p) ends up owning the dd = kh corner quad of its column on
//     ALL levels: every level's entry stores are full-wave instructions, 4 per lane;
//   * per level the box of corner voxels (out-of-level voxels are staged as zero rows: no per-corner
//     validity) is walked in K-blocks of 32 rows: global -> registers -> LDS as whole 128-byte head slices,
//     prefetched a block ahead; per 16 rows: A = the lane's own weight column (2 ds_read_b128, 8 v_perm to
//     separate hi from lo), B = V read by the transposing ds_read_b64_tr_b16, 4 v_mfma_f32_32x32x16;
//   * D comes out [(query, point)][channel]: the four points of a query are four registers of one lane --
//     3 adds, then the 8 output rows leave through LDS as whole 128-byte lines;
//   * a level whose box exceeds kPcmBoxRows rows (non-local sampling) uses the same loop over an explicit
//     row list: one K-slot per (column, corner), 256 per level.
#pragma once
#include "msda3d_common.hpp"
#include "msda3d_mma.hpp"

namespace transoar {

constexpr int kPcmKB = 32;                // value rows per K-block
constexpr int kPcmVP = 128;               // bytes per staged row, no padding: the 16-byte pieces of a row are stored XORed with 4 * ((row >> 1) & 1),
                                          // which makes the four 64-byte windows a transposing read takes out of 4 consecutive rows tile the 64 banks
constexpr int kPcmKW = 32;                // K-slots of a weight column: 32 (one staged block of rows) or 64 (two)
constexpr int kPcmWP = kPcmKW + 4;        // dwords per weight column: 64 K-slots + 4 spares (one is used, by column: bank spread) for entries outside the block
constexpr int kPcmBoxRows = 256;          // larger boxes: explicit (column, corner) slots instead (also 256 rows)
constexpr int kPcmLevels = 4;
constexpr int kPcmPU = 272, kPcmPF = 136;   // bytes per query of the staged parameter block (loc + attn fp32 / projection bf16), padded: the 8 queries start in different banks
constexpr int kPcmSkip = 0x3fffffff;      // packed (d0, h0, w0) of a point that is skipped

template <typename VT> struct PcmW;       // one trilinear weight -> (hi << 16) | lo, two 16-bit terms in the storage type
template <> struct PcmW<bf16_t> {
  static __device__ __forceinline__ unsigned split(float w) {
    const unsigned u = __float_as_uint(w);
    const float lo = w - __uint_as_float(u & 0xffff0000u);          // exact
    return __builtin_amdgcn_perm(u, __float_as_uint(lo), 0x07060302u);
  }
  static __device__ __forceinline__ unsigned pack2(float a, float b) {      // (bf16(b) << 16) | bf16(a), round to nearest even
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
  }
};
template <> struct PcmW<f16_t> {
  static __device__ __forceinline__ unsigned split(float w) {
    const _Float16 hi = static_cast<_Float16>(w);
    const _Float16 lo = static_cast<_Float16>(w - static_cast<float>(hi));
    return (static_cast<unsigned>(__builtin_bit_cast(unsigned short, hi)) << 16) | __builtin_bit_cast(unsigned short, lo);
  }
  static __device__ __forceinline__ unsigned pack2(float a, float b) {
    return (static_cast<unsigned>(__builtin_bit_cast(unsigned short, static_cast<_Float16>(b))) << 16) |
           __builtin_bit_cast(unsigned short, static_cast<_Float16>(a));
  }
};

// A copy of v the optimiser cannot identify with v.  hipcc (ROCm 7.2) folds v_permlane{16,32}_swap of two identical
// operands into a no-op -- the two results are different permutations of the lanes, not (v, v).
__device__ __forceinline__ unsigned opaque_copy(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}

// minimum of two packed int16 over the 32 lanes of each wave half; the result is in the half's LAST lane (31 / 63)
__device__ __forceinline__ int half_min_pk16(int v) {
#define TRANSOAR_PCM_STEP(CTRL)                                                                  \
  {                                                                                              \
    const int o = __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);                             \
    v = __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(s16x2, v), __builtin_bit_cast(s16x2, o))); \
  }
  TRANSOAR_PCM_STEP(0xB1)     // quad_perm [1,0,3,2]
  TRANSOAR_PCM_STEP(0x4E)     // quad_perm [2,3,0,1]
  TRANSOAR_PCM_STEP(0x141)    // row_half_mirror
  TRANSOAR_PCM_STEP(0x140)    // row_mirror: every lane holds its 16-lane row's minimum
#undef TRANSOAR_PCM_STEP
  const int o = __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false);      // row_bcast15 into rows 1 and 3
  return __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(s16x2, v), __builtin_bit_cast(s16x2, o)));
}
template <int CTRL>
__device__ __forceinline__ float quad_perm_f(float v) {                // value of the lane at quad_perm CTRL
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
// both halves' values of v: {value of lane & 31, value of (lane & 31) + 32}
__device__ __forceinline__ void both_halves_f(float v, float& lo, float& hi) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), opaque_copy(__float_as_uint(v)), false, false);
  lo = __uint_as_float(sw[0]);
  hi = __uint_as_float(sw[1]);
}

__device__ __forceinline__ float bf16_round_f(float x) {      // round-to-nearest-even to bf16 (finite x), as a float
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return __uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// pixel coordinate loc * size - 0.5, multiply and subtract rounded separately like the scalar oracle (no FMA)
__device__ __forceinline__ float pixel_coord_f(float loc, float size) {
#pragma clang fp contract(off)
  const float t = loc * size;
  return t - 0.5f;
}

struct PcmBox {         // wave-uniform: box of corner voxels of one level (may reach one voxel outside the level)
  int bd, bh, bw, TD, TH, TW;
};

// Launch constants of the point-column gather (device memory, read through scalar loads)
struct PcmConst {
  BrickOrder order;
  float fD[4], fH[4], fW[4];          // level sizes as floats
  float dD[4], dH[4], dW[4];          // FUSED: what the autocast chain divides the offsets by: bf16(size)
};

template <typename VT, bool FUSED>
__global__ __launch_bounds__(64, kPcmKW == 32 ? 4 : 3) void msda3d_fwd_pcm(
    const VT* __restrict__ value, const float* __restrict__ loc, const float* __restrict__ attn,
    const unsigned short* __restrict__ proj, const float* __restrict__ ref, unsigned ref_bstride,
    VT* __restrict__ out, int S, int M, int L, unsigned value_bytes, unsigned param_bytes, unsigned aux_bytes,
    unsigned n_units, const PcmConst* __restrict__ cst) {
  const BrickOrder& order = cst->order;
  constexpr int C = 64, KB = kPcmKB, KW = kPcmKW, VP = kPcmVP, WP = kPcmWP;
  __shared__ __attribute__((aligned(128))) unsigned char vbuf[KB * VP];       // staged rows; parameter block and output rows alias it
  __shared__ __attribute__((aligned(16))) unsigned wbuf[32 * WP];            // [column][K-slot]: (hi << 16) | lo

  // XCD-contiguous work order (block b runs on XCD b % 8: each XCD walks one contiguous eighth)
  const unsigned per_xcd = (n_units + 7u) >> 3;
  const unsigned u = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (u >= n_units) return;
  const int lane = threadIdx.x;
  const int kh = lane >> 5, n = lane & 31, q = n >> 2, p = n & 3;
  const unsigned sub = u & 15u;
  const unsigned t1 = u >> 4;
  const unsigned m = t1 % static_cast<unsigned>(M);
  const unsigned t2 = t1 / static_cast<unsigned>(M);
  const unsigned bricks = static_cast<unsigned>(order.pad_start[order.L]) >> 7;
  const int brick = static_cast<int>(bricks - 1u - t2 % bricks);        // coarse levels first: their boxes are the big ones
  const unsigned b = t2 / bricks;

  // ---- the wave's 2x2x2 queries: level of the brick, its origin, the sub-brick's origin (all wave-uniform)
  int lq = 0;
#pragma unroll
  for (int t = 1; t < kPcmLevels; ++t) lq += (t < order.L && brick * kBrickSlots >= order.pad_start[t]) ? 1 : 0;
  const int qD = order.D[lq], qH = order.H[lq], qW = order.W[lq];
  const unsigned lbrick = static_cast<unsigned>(brick - (order.pad_start[lq] >> 7));
  const unsigned nbw = static_cast<unsigned>(order.nbw[lq]), nbh = static_cast<unsigned>(order.nbh[lq]);
  const unsigned bwi = lbrick % nbw, brest = lbrick / nbw;
  const unsigned bhi = brest % nbh, bdi = brest / nbh;
  const int od = static_cast<int>(bdi * kBrickD + 2 * (sub >> 3)), oh = static_cast<int>(bhi * kBrickH + 2 * ((sub >> 2) & 1)),
            ow = static_cast<int>(bwi * kBrickW + 2 * (sub & 3));
  if (od >= qD || oh >= qH || ow >= qW) return;                         // the whole sub-brick is padding
  const int qbase = order.start[lq] + static_cast<int>(b) * S;
  auto row_of = [&](int qq) -> int {                                     // b * S + pyramid row of query qq of the wave, -1 = padding
    const int d = od + (qq >> 2), h = oh + ((qq >> 1) & 1), w = ow + (qq & 1);
    const int r = qbase + __mul24(__mul24(d, qH) + h, qW) + w;
    return (d < qD && h < qH && w < qW) ? r : -1;
  };
  const int s = row_of(q);
  const bool live = s >= 0;
  const unsigned row_bytes = static_cast<unsigned>(M) * C * sizeof(VT);
  const unsigned head_off = (b * static_cast<unsigned>(S) * M + m) * (C * sizeof(VT));
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<VT*>(value), 0, static_cast<int>(value_bytes), 0x00020000);

  // ---- parameter block of the 8 (query, head) items -> LDS: 4 queries x 16 pieces per round; pieces 0..11 of a
  // query are its locations / offsets (3 per level), pieces 12..15 its weights / logits (1 per level)
  {
    const int r = lane & 15;
    const bool loc_piece = r < 12;
    const int lv = loc_piece ? r : 3 * (r - 12);
    const bool on = lv < 3 * L;
    if constexpr (FUSED) {
      const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(proj), 0, static_cast<int>(param_bytes), 0x00020000);
      const unsigned cols2 = 8u * M * L * 4;                           // bytes per projection row: 4 * M * L * P bf16
      const unsigned in_row = loc_piece ? (m * L * 12 + r * 4) * 2u : (3u * M * L * 4 + m * L * 4 + (r - 12) * 4) * 2u;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int qq = (lane >> 4) + 4 * it;
        const int sq = row_of(qq);
        const unsigned off = (on && sq >= 0) ? __umul24(static_cast<unsigned>(sq), cols2) + in_row : 0xfffffff0u;
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t v = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(prs, off, 0, 0));
        *reinterpret_cast<u32x2_t*>(vbuf + qq * kPcmPF + r * 8) = v;
      }
    } else {
      const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(loc), 0, static_cast<int>(param_bytes), 0x00020000);
      const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(attn), 0, static_cast<int>(aux_bytes), 0x00020000);
      const unsigned item_loc = static_cast<unsigned>(L) * 48u, item_attn = static_cast<unsigned>(L) * 16u;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int qq = (lane >> 4) + 4 * it;
        const int sq = row_of(qq);
        const unsigned item = __umul24(static_cast<unsigned>(sq), static_cast<unsigned>(M)) + m;
        const bool ok = on && sq >= 0;
        const unsigned loff = (ok && loc_piece) ? item * item_loc + r * 16 : 0xfffffff0u;
        const unsigned aoff = (ok && !loc_piece) ? item * item_attn + (r - 12) * 16 : 0xfffffff0u;
        // (aux = 2: the non-temporal hint.  Locations, weights and the output rows pass through once; without the hint
        // they push value rows out of the XCD's L2: HBM fetch 512 -> 457 MB per launch, round 5)
        const u32x4 vl = __builtin_amdgcn_raw_buffer_load_b128(lrs, loff, 0, 2);
        const u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(ars, aoff, 0, 2);
        *reinterpret_cast<u32x4*>(vbuf + qq * kPcmPU + r * 16) = loc_piece ? vl : va;
      }
    }
  }

  // ---- geometry of this lane's two points: (level 2kh, point p) and (level 2kh + 1, point p) of query q
  float fl[2][3], fa[2];            // fractional parts (d, h, w) and attention weight
  int dhw[2];                       // (d0 + 1) | (h0 + 1) << 10 | (w0 + 1) << 20, kPcmSkip for a skipped point
  int mn[3], mx[3];                 // packed (point 0, point 1) minima of d0 / h0 / w0 and of their negatives
  {
    float px[2][3];                 // normalised location (x, y, z)
    float fs[2][3];                 // (W, H, D) of the point's level
    bool okl[2];
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      okl[pi] = live && 2 * kh + pi < L;
      fs[pi][0] = kh ? cst->fW[2 + pi] : cst->fW[pi];
      fs[pi][1] = kh ? cst->fH[2 + pi] : cst->fH[pi];
      fs[pi][2] = kh ? cst->fD[2 + pi] : cst->fD[pi];
    }
    if constexpr (FUSED) {
      const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ref), 0, static_cast<int>(aux_bytes), 0x00020000);
      const unsigned short* pb = reinterpret_cast<const unsigned short*>(vbuf + q * kPcmPF);
      const unsigned rrow = live ? (static_cast<unsigned>(s) - b * static_cast<unsigned>(S)) * L * 12u + b * ref_bstride : 0xfffffff0u;
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const int l = 2 * kh + pi;
        typedef float f32x3_t __attribute__((ext_vector_type(3)));
        const f32x3_t rp = __builtin_bit_cast(f32x3_t, __builtin_amdgcn_raw_buffer_load_b96(rrs, okl[pi] ? rrow + l * 12u : 0xfffffff0u, 0, 0));
        const float dv[3] = {kh ? cst->dW[2 + pi] : cst->dW[pi], kh ? cst->dH[2 + pi] : cst->dH[pi], kh ? cst->dD[2 + pi] : cst->dD[pi]};
        fa[pi] = okl[pi] ? bf16_to_f32(pb[48 + l * 4 + p]) : -3.0e38f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float off = bf16_to_f32(pb[(l * 4 + p) * 3 + k]);
          // off / dv correctly rounded: quotient estimate + one residual step (dv is a small integer: no scaling needed)
          const float rc = __builtin_amdgcn_rcpf(dv[k]);
          const float q0 = off * rc;
          const float qv = __builtin_fmaf(__builtin_fmaf(-q0, dv[k], off), rc, q0);
          px[pi][k] = rp[k] + bf16_round_f(qv);
        }
      }
      // softmax over the L*P logits of (q, head): 2 in this lane, x 4 points (quad) x 2 halves
      float mxl = fmaxf(fa[0], fa[1]);
      mxl = fmaxf(mxl, quad_perm_f<0xB1>(mxl));
      mxl = fmaxf(mxl, quad_perm_f<0x4E>(mxl));
      {
        float a0, a1;
        both_halves_f(mxl, a0, a1);
        mxl = fmaxf(a0, a1);
      }
      const float e0 = __expf(fa[0] - mxl), e1 = __expf(fa[1] - mxl);
      float sum = e0 + e1;
      sum += quad_perm_f<0xB1>(sum);
      sum += quad_perm_f<0x4E>(sum);
      {
        float a0, a1;
        both_halves_f(sum, a0, a1);
        sum = a0 + a1;
      }
      fa[0] = e0 / sum; fa[1] = e1 / sum;
    } else {
      const float* pb = reinterpret_cast<const float*>(vbuf + q * kPcmPU);
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const int l = 2 * kh + pi;
        px[pi][0] = pb[(l * 4 + p) * 3]; px[pi][1] = pb[(l * 4 + p) * 3 + 1]; px[pi][2] = pb[(l * 4 + p) * 3 + 2];
        fa[pi] = pb[48 + l * 4 + p];
      }
    }
    int i0[2][3];
    unsigned okmask = 0u;
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const float w_im = pixel_coord_f(px[pi][0], fs[pi][0]), h_im = pixel_coord_f(px[pi][1], fs[pi][1]), d_im = pixel_coord_f(px[pi][2], fs[pi][2]);
      const bool ok = okl[pi] && d_im > -1.f && h_im > -1.f && w_im > -1.f && d_im < fs[pi][2] && h_im < fs[pi][1] && w_im < fs[pi][0];
      const float fd = floorf(d_im), fh = floorf(h_im), fw = floorf(w_im);
      i0[pi][0] = static_cast<int>(fd); i0[pi][1] = static_cast<int>(fh); i0[pi][2] = static_cast<int>(fw);
      fl[pi][0] = d_im - fd; fl[pi][1] = h_im - fh; fl[pi][2] = w_im - fw;
      fa[pi] = ok ? fa[pi] : 0.f;
      dhw[pi] = ok ? (i0[pi][0] + 1) | ((i0[pi][1] + 1) << 10) | ((i0[pi][2] + 1) << 20) : kPcmSkip;
      okmask |= ok ? (pi ? 0xffff0000u : 0x0000ffffu) : 0u;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned pk = __builtin_amdgcn_perm(static_cast<unsigned>(i0[1][k]), static_cast<unsigned>(i0[0][k]), 0x05040100u);
      // highest corner voxel a point really touches along this axis: i0 + 1 only if its fraction is non-zero.  On integer
      // pixel coordinates (the module's initial state: offsets of whole voxels, ms_deform_attn.py:67-82) seven of the
      // eight corners have weight zero; they then do not widen the box (SURVEY appendix A's fast path) -- their entries
      // go to the spare slot below
      const int t0 = i0[0][k] + (fl[0][k == 0 ? 0 : k == 1 ? 1 : 2] > 0.f ? 1 : 0), t1 = i0[1][k] + (fl[1][k == 0 ? 0 : k == 1 ? 1 : 2] > 0.f ? 1 : 0);
      const unsigned tp = __builtin_amdgcn_perm(static_cast<unsigned>(t1), static_cast<unsigned>(t0), 0x05040100u);
      const unsigned ng = __builtin_bit_cast(unsigned, -__builtin_bit_cast(s16x2, tp));
      // a skipped point is neutral (32767) in both
      mn[k] = half_min_pk16(static_cast<int>((pk & okmask) | (0x7fff7fffu & ~okmask)));
      mx[k] = half_min_pk16(static_cast<int>((ng & okmask) | (0x7fff7fffu & ~okmask)));
    }
  }

  // ---- boxes of the four levels (wave-uniform): lane 31 holds levels (0, 1), lane 63 levels (2, 3)
  PcmBox box[kPcmLevels];
  int mode[kPcmLevels];             // 0 = no valid point, 1 = box rows, 2 = explicit (column, corner) slots
#pragma unroll
  for (int l = 0; l < kPcmLevels; ++l) {
    const int src = (l >> 1) * 32 + 31;
    int lo3[3], hi3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int a = __builtin_amdgcn_readlane(mn[k], src), c = __builtin_amdgcn_readlane(mx[k], src);
      lo3[k] = (l & 1) ? (a >> 16) : static_cast<short>(a);
      hi3[k] = -((l & 1) ? (c >> 16) : static_cast<int>(static_cast<short>(c)));
    }
    box[l] = PcmBox{lo3[0], lo3[1], lo3[2], hi3[0] - lo3[0] + 1, hi3[1] - lo3[1] + 1, hi3[2] - lo3[2] + 1};
    mode[l] = (l >= L || lo3[0] == 32767) ? 0 : (box[l].TD * box[l].TH * box[l].TW <= kPcmBoxRows ? 1 : 2);
  }

  // ---- weight entries of the lane's two points, split into hi | lo; then the halves trade quads:
  // afterwards wq[l][dh * 2 + dw] is the (dd = kh) corner quad of column n on level l, pdhw[l] the column's point there
  unsigned wq[kPcmLevels][4];
  int pdhw[kPcmLevels];
  {
    unsigned X[2][4], Y[2][4];
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const float ld = fl[pi][0], lh = fl[pi][1], lw = fl[pi][2];
      const float wd[2] = {fa[pi] * (1.f - ld), fa[pi] * ld}, wh[2] = {1.f - lh, lh}, ww[2] = {1.f - lw, lw};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float f0 = wd[0] * wh[c >> 1], f1 = wd[1] * wh[c >> 1];
        X[pi][c] = PcmW<VT>::split(f0 * ww[c & 1]);
        Y[pi][c] = PcmW<VT>::split(f1 * ww[c & 1]);
      }
    }
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // X: lanes 32..63 <-> Y: lanes 0..31.  After it X holds level pi's quad for every lane, Y level 2 + pi's
        const auto sw = __builtin_amdgcn_permlane32_swap(X[pi][c], Y[pi][c], false, false);
        wq[pi][c] = sw[0];
        wq[2 + pi][c] = sw[1];
      }
      const auto sd = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(dhw[pi]), opaque_copy(static_cast<unsigned>(dhw[pi])), false, false);
      pdhw[pi] = static_cast<int>(sd[0]);
      pdhw[2 + pi] = static_cast<int>(sd[1]);
    }
  }

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  for (int i = lane; i < 32 * WP / 2; i += 64) reinterpret_cast<uint2*>(wbuf)[i] = uint2{0u, 0u};

  // Byte offsets of the 64 rows [k0, k0 + 64) of level l (lane i: row k0 + i); a row past the end or outside the
  // level gets an offset past the buffer: it must read as zeros in ALL its 16-byte pieces (the piece offset is
  // added later), hence 0xffffff00 and not the usual 0xfffffff0.
  auto row_offsets = [&](auto lc, int k0) -> int {
    constexpr int l = decltype(lc)::value;
    const PcmBox bx = box[l];
    const int D = order.D[l], H = order.H[l], W = order.W[l], start = order.start[l];
    const int r = k0 + lane;
    int d, h, w;
    bool ok;
    if (mode[l] == 1) {
      const int THW = bx.TH * bx.TW, R = bx.TD * THW;
      // r -> (rd, rh, rw) by float reciprocals: (r + 0.5) / n is >= 0.5 / n away from an integer, far more than
      // the float error for r < 2^12
      const float inv_thw = __builtin_amdgcn_rcpf(static_cast<float>(THW)), inv_tw = __builtin_amdgcn_rcpf(static_cast<float>(bx.TW));
      const int rd = static_cast<int>((static_cast<float>(r) + 0.5f) * inv_thw), rr = r - __mul24(rd, THW);
      const int rh = static_cast<int>((static_cast<float>(rr) + 0.5f) * inv_tw), rw = rr - __mul24(rh, bx.TW);
      d = bx.bd + rd; h = bx.bh + rh; w = bx.bw + rw;
      ok = r < R;
    } else {
      // explicit slots: slot r = column (r >> 3), corner (r & 7) = dd*4 + dh*2 + dw
      const int pd = __builtin_amdgcn_ds_bpermute((r >> 3) * 4, pdhw[l]);
      d = (pd & 1023) - 1 + ((r >> 2) & 1); h = ((pd >> 10) & 1023) - 1 + ((r >> 1) & 1); w = ((pd >> 20) & 1023) - 1 + (r & 1);
      ok = pd != kPcmSkip;
    }
    ok = ok && static_cast<unsigned>(d) < static_cast<unsigned>(D) && static_cast<unsigned>(h) < static_cast<unsigned>(H) &&
         static_cast<unsigned>(w) < static_cast<unsigned>(W);
    const int grow = start + __mul24(__mul24(d, H) + h, W) + w;
    return ok ? static_cast<int>(head_off + __umul24(static_cast<unsigned>(grow), row_bytes)) : static_cast<int>(0xffffff00u);
  };
  // the 32 rows of one half of the 64: 4 x 16 bytes per lane; the 8 lanes that fetch a row get its offset through ds_bpermute
  const int st_row = lane >> 3, st_vec = lane & 7;
  const int st_swz = (st_vec ^ ((st_row & 2) << 1)) * 16;         // rows it * 8 + st_row: bit 1 of the row = bit 1 of st_row
  auto issue_loads = [&](int row_off, int half, u32x4 (&pre)[4]) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const unsigned off = static_cast<unsigned>(__builtin_amdgcn_ds_bpermute((half * 32 + it * 8 + st_row) * 4, row_off)) + st_vec * 16u;
      pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    }
  };

  u32x4 pre[4];
  bool have_pre = false;
  int row_off = 0;
  // byte offsets of the lane's two transposing reads inside a staged row (channel tiles 0 and 1), swizzled like the stores
  const int tr_off0 = ((16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2) ^ ((((lane & 15) >> 2) & 2) << 5), tr_off1 = tr_off0 ^ 64;
  unsigned* const wcol = wbuf + n * WP;
  const int spare = KW + (n >> 3);            // the column's spare slot (entries outside the block): the 4 slots past the K-slots, so that the 32 columns' spares sit in 32 different banks
  static_for<0, kPcmLevels>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    if (mode[l] == 0) return;
    const PcmBox bx = box[l];
    const int THW = bx.TH * bx.TW;
    const int R = mode[l] == 1 ? bx.TD * THW : 256;
    // K-slots of the lane's four entries on this level
    int col[4];
    if (mode[l] == 1) {
      const int pd = pdhw[l];
      const int d1 = pd & 1023, h1 = (pd >> 10) & 1023, w1 = pd >> 20;        // d0 + 1, h0 + 1, w0 + 1
      const int base = __mul24(__mul24(bx.bd + 1, bx.TH) + (bx.bh + 1), bx.TW) + (bx.bw + 1);
      const int c0 = pd == kPcmSkip ? 0 : __mul24(__mul24(d1 + kh, bx.TH) + h1, bx.TW) + w1 - base;
      col[0] = c0; col[1] = c0 + 1; col[2] = c0 + bx.TW; col[3] = c0 + bx.TW + 1;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) col[c] = n * 8 + kh * 4 + c;
    }
    const int nh = (R + KB - 1) / KB;                                     // halves of 32 rows
    if (!have_pre) {
      row_off = row_offsets(lc, 0);
      issue_loads(row_off, 0, pre);
    }
    int wad[4];
    for (int hb = 0; hb < nh; ++hb) {
      const int half = hb & 1;                                            // which half of the 64 row offsets in row_off
      const int sub = hb & (KW / KB - 1);                                 // which staged block of the weight column's K-slots
      const int nch = (min(KB, R - hb * KB) + 15) >> 4;                   // 16-row chunks of this half: 1 or 2
      // ---- staged rows -> LDS
#pragma unroll
      for (int it = 0; it < 2; ++it) *reinterpret_cast<u32x4*>(vbuf + (it * 8 + st_row) * VP + st_swz) = pre[it];
      if (nch > 1) {
#pragma unroll
        for (int it = 2; it < 4; ++it) *reinterpret_cast<u32x4*>(vbuf + (it * 8 + st_row) * VP + st_swz) = pre[it];
      }
      // ---- the lane's four entries for the 64 rows that start here: slot inside the block, or the spare slot
      if (sub == 0) {
        const int k0 = hb * KB;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned t = static_cast<unsigned>(col[c] - k0);
          // a zero weight (a corner whose fraction is zero, a skipped point) may lie outside the box: never into a real slot
          wad[c] = (t < static_cast<unsigned>(KW) && wq[l][c] != 0u) ? static_cast<int>(t) : spare;
          wcol[wad[c]] = wq[l][c];
        }
      }
      // ---- prefetch the next half (of this level, or the first of the next one)
      have_pre = false;
      if (hb + 1 < nh) {
        if (half == 1) row_off = row_offsets(lc, (hb + 1) * KB);
        issue_loads(row_off, half ^ 1, pre);
        have_pre = true;
      } else if constexpr (l + 1 < kPcmLevels) {
        if (mode[l + 1] != 0) {
          row_off = row_offsets(IntC<l + 1>{}, 0);
          issue_loads(row_off, 0, pre);
          have_pre = true;
        }
      }
      // ---- 16-row chunks on the matrix cores (unrolled: the second chunk's reads are the first one's at immediate offsets --
      // as a run-time loop three address registers were stepped per chunk, 48 vector instructions per wave)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        if (kc >= nch) break;
        const unsigned* wp = wcol + sub * KB + kc * 16 + 8 * kh;
        const u32x4 p0 = *reinterpret_cast<const u32x4*>(wp);
        const u32x4 p1 = *reinterpret_cast<const u32x4*>(wp + 4);
        const u32x4 ahi{__builtin_amdgcn_perm(p0[1], p0[0], 0x07060302u), __builtin_amdgcn_perm(p0[3], p0[2], 0x07060302u),
                        __builtin_amdgcn_perm(p1[1], p1[0], 0x07060302u), __builtin_amdgcn_perm(p1[3], p1[2], 0x07060302u)};
        const u32x4 alo{__builtin_amdgcn_perm(p0[1], p0[0], 0x05040100u), __builtin_amdgcn_perm(p0[3], p0[2], 0x05040100u),
                        __builtin_amdgcn_perm(p1[1], p1[0], 0x05040100u), __builtin_amdgcn_perm(p1[3], p1[2], 0x05040100u)};
        // B = V: lane supplies row (lane & 15) >> 2 of its group's 4-row set, 4 channels; receives its channel's column
        // rows kc*16 + 8 kh + r (and + 4), r = (lane & 15) >> 2: bit 1 of the row is bit 1 of r -> the lane's swizzle is constant
        const unsigned char* vrow = vbuf + (kc * 16 + 8 * kh + ((lane & 15) >> 2)) * VP;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const s16x4 b00 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vrow + tr_off0));
        const s16x4 b01 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vrow + 4 * VP + tr_off0));
        const s16x4 b10 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vrow + tr_off1));
        const s16x4 b11 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vrow + 4 * VP + tr_off1));
        const s16x8 v0 = __builtin_shufflevector(b00, b01, 0, 1, 2, 3, 4, 5, 6, 7);
        const s16x8 v1 = __builtin_shufflevector(b10, b11, 0, 1, 2, 3, 4, 5, 6, 7);
        const s16x8 whi = __builtin_bit_cast(s16x8, ahi), wlo = __builtin_bit_cast(s16x8, alo);
        acc0 = Mma<VT>::mfma(whi, v0, acc0);
        acc1 = Mma<VT>::mfma(whi, v1, acc1);
        acc0 = Mma<VT>::mfma(wlo, v0, acc0);
        acc1 = Mma<VT>::mfma(wlo, v1, acc1);
      }
      // ---- clear the entries again when their 64 rows are done
      if (sub == KW / KB - 1 || hb + 1 == nh) {
#pragma unroll
        for (int c = 0; c < 4; ++c) wcol[wad[c]] = 0u;
      }
    }
  });

  // ---- D[(query, point)][channel]: register r of a lane is column (r & 3) + 8 (r >> 2) + 4 kh = query 2 (r >> 2) + kh,
  // point r & 3, channel lane & 31 (+ 32 for the second tile).  Sum the points, rows leave through LDS.
  {
    unsigned short* ob = reinterpret_cast<unsigned short*>(vbuf);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float v0 = (acc0[4 * g] + acc0[4 * g + 1]) + (acc0[4 * g + 2] + acc0[4 * g + 3]);
      const float v1 = (acc1[4 * g] + acc1[4 * g + 1]) + (acc1[4 * g + 2] + acc1[4 * g + 3]);
      const unsigned pk = PcmW<VT>::pack2(v0, v1);
      ob[(2 * g + kh) * C + n] = static_cast<unsigned short>(pk);
      ob[(2 * g + kh) * C + 32 + n] = static_cast<unsigned short>(pk >> 16);
    }
    const int sq = row_of(lane >> 3);
    const u32x4 line = *reinterpret_cast<const u32x4*>(vbuf + (lane >> 3) * 128 + (lane & 7) * 16);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(out, 0, static_cast<int>(value_bytes), 0x00020000);
    const unsigned ooff = sq >= 0 ? (__umul24(static_cast<unsigned>(sq), static_cast<unsigned>(M)) + m) * (C * sizeof(VT)) + (lane & 7) * 16u : 0xfffffff0u;
    __builtin_amdgcn_raw_buffer_store_b128(line, ors, ooff, 0, 2);
  }
}

}  // namespace transoar
