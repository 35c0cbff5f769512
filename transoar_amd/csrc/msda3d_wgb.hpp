// Forward gather of MSDeformAttn-3D on the matrix cores, workgroup-box form (gfx950) -- round 5.
//
// Semantics: SURVEY.md appendix A (ops/src/cuda/ms_deform_im2col_cuda.cuh:31-114, 370-439); with FUSED the
// sampling head of the module as well (ops/modules/ms_deform_attn.py:114-128).
//
// msda3d_pcm.hpp (round 3) gives every wave its own box of value rows, fetched global -> registers -> LDS 32 rows
// at a time with one block of prefetch: at the flagship size the 175 500 waves ask the L2 for 5.7 GB of rows (32x
// the tensor), a wave is parked on a load or an LDS store for 60 % of its life, and the counters of two rounds said
// so (WAIT_ANY 27 %, WAIT_INST 33 % of the wave cycles, 16 % of the HBM roofline).  Same matrix-core mathematics
// here (32 MFMA columns = 8 queries x 4 points, a column holds one point per level, entries written never
// accumulated, bf16 hi | lo split), different data movement:
//
//   * a WORKGROUP of 4 waves owns a 4 x 2 x 4 (d, h, w) block of queries of one head; wave i owns the d-plane i
//     (2 x 4 queries).  The waves exchange their per-level corner boxes through LDS and take the UNION box;
//   * the union box of every level is staged ONCE per workgroup, by LDS-DMA (buffer_load_dwordx4 ... lds: no
//     registers, no LDS store instructions), all levels of a round in flight together -- one exposed latency per
//     round instead of one per 32 rows; a union that does not fit the row buffer is staged d-slab by d-slab over
//     several rounds (the MFMA accumulators simply carry on);
//   * rows are stored in union order (d, h, w): the rows a wave needs -- its own d-range, the union's (h, w)
//     extent -- are one contiguous K range of the buffer, so the MFMA loop addresses them like msda3d_pcm.hpp
//     addressed its private block; a 32-row block none of the wave's 128 entries falls into is skipped;
//   * a level whose boxes are too large (non-local sampling) runs on explicit (column, corner) rows, 32 at a time
//     per wave, double-buffered in the wave's quarter of the row buffer.
#pragma once
#include "msda3d_pcm.hpp"

namespace transoar {

constexpr int kWgbWaves = 4;
constexpr int kWgbRows = 272;             // capacity of the staged-row buffer (rows of 128 bytes): 34 KB; with the weight blocks 53.5 KB per workgroup = 3 per CU
constexpr int kWgbPad = 24;               // zero rows after a segment: the last 16-row chunk of a wave may run 15 rows (+ 3 of alignment) past it
constexpr int kWgbMaxK = 256;             // box mode only while a wave's K range stays within this many rows

typedef __attribute__((address_space(3))) void wgb_lds_void;

// 64 lanes x 16 bytes, global -> LDS, lane i lands at LDS byte dst + 16 i (dst wave-uniform).  Inline assembly for the
// reason given in mfma_stream.hpp: hipcc neither counts these loads nor waits for them (wgb_dma_wait does).
__device__ __forceinline__ void wgb_dma(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned dst) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rs), "s"(dst)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wgb_dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename VT, bool FUSED>
__global__ __launch_bounds__(64 * kWgbWaves, 3) void msda3d_fwd_wgb(
    const VT* __restrict__ value, const float* __restrict__ loc, const float* __restrict__ attn,
    const unsigned short* __restrict__ proj, const float* __restrict__ ref, unsigned ref_bstride,
    VT* __restrict__ out, int S, int M, int L, unsigned value_bytes, unsigned param_bytes, unsigned aux_bytes,
    unsigned n_units, const PcmConst* __restrict__ cst) {
  const BrickOrder& order = cst->order;
  constexpr int C = 64, KB = kPcmKB, KW = kPcmKW, VP = kPcmVP, WP = kPcmWP;
  static_assert(KW == 32 && KB == 32, "one staged block of rows per weight window");
  __shared__ __attribute__((aligned(128))) unsigned char rbuf[kWgbRows * VP];     // staged rows, union order
  __shared__ __attribute__((aligned(16))) unsigned wbuf_all[kWgbWaves * 32 * WP];   // per wave [column][K-slot]: (hi << 16) | lo; parameter block and output rows alias it
  __shared__ int ubox[kWgbWaves][16];

  // XCD-contiguous work order (block b runs on XCD b % 8: each XCD walks one contiguous eighth)
  const unsigned per_xcd = (n_units + 7u) >> 3;
  const unsigned u = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (u >= n_units) return;
  const int lane = threadIdx.x & 63;
  const int wave = sgpr(static_cast<int>(threadIdx.x >> 6));
  const int kh = lane >> 5, n = lane & 31, q = n >> 2, p = n & 3;
  const unsigned quarter = u & 3u;
  const unsigned t1 = u >> 2;
  const unsigned m = t1 % static_cast<unsigned>(M);
  const unsigned t2 = t1 / static_cast<unsigned>(M);
  const unsigned bricks = static_cast<unsigned>(order.pad_start[order.L]) >> 7;
  const int brick = static_cast<int>(bricks - 1u - t2 % bricks);        // coarse levels first: their boxes are the big ones
  const unsigned b = t2 / bricks;

  // ---- the workgroup's 4x2x4 queries: level of the brick, its origin, the quarter's origin (all uniform)
  int lq = 0;
#pragma unroll
  for (int t = 1; t < kPcmLevels; ++t) lq += (t < order.L && brick * kBrickSlots >= order.pad_start[t]) ? 1 : 0;
  const int qD = order.D[lq], qH = order.H[lq], qW = order.W[lq];
  const unsigned lbrick = static_cast<unsigned>(brick - (order.pad_start[lq] >> 7));
  const unsigned nbw = static_cast<unsigned>(order.nbw[lq]), nbh = static_cast<unsigned>(order.nbh[lq]);
  const unsigned bwi = lbrick % nbw, brest = lbrick / nbw;
  const unsigned bhi = brest % nbh, bdi = brest / nbh;
  const int od = static_cast<int>(bdi * kBrickD) + wave, oh = static_cast<int>(bhi * kBrickH + 2 * (quarter >> 1)),
            ow = static_cast<int>(bwi * kBrickW + 4 * (quarter & 1));
  if (oh >= qH || ow >= qW) return;                                      // the whole quarter is padding (every wave agrees)
  const int qbase = order.start[lq] + static_cast<int>(b) * S;
  auto row_of = [&](int qq) -> int {                                     // b * S + pyramid row of query qq of the wave, -1 = padding
    const int h = oh + (qq >> 2), w = ow + (qq & 3);
    const int r = qbase + __mul24(__mul24(od, qH) + h, qW) + w;
    return (od < qD && h < qH && w < qW) ? r : -1;
  };
  const int s = row_of(q);
  const bool live = s >= 0;
  const unsigned row_bytes = static_cast<unsigned>(M) * C * sizeof(VT);
  const unsigned head_off = (b * static_cast<unsigned>(S) * M + m) * (C * sizeof(VT));
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<VT*>(value), 0, static_cast<int>(value_bytes), 0x00020000);
  unsigned* const wbuf = wbuf_all + wave * (32 * WP);
  unsigned char* const pblk = reinterpret_cast<unsigned char*>(wbuf);    // parameter block of the wave's 8 (query, head) items

  // ---- parameter block -> LDS: 4 queries x 16 pieces per round; pieces 0..11 of a query are its locations /
  // offsets (3 per level), pieces 12..15 its weights / logits (1 per level)
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
        *reinterpret_cast<u32x2_t*>(pblk + qq * kPcmPF + r * 8) = v;
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
        const u32x4 vl = __builtin_amdgcn_raw_buffer_load_b128(lrs, loff, 0, 0);
        const u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(ars, aoff, 0, 0);
        *reinterpret_cast<u32x4*>(pblk + qq * kPcmPU + r * 16) = loc_piece ? vl : va;
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
      const unsigned short* pb = reinterpret_cast<const unsigned short*>(pblk + q * kPcmPF);
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
      const float* pb = reinterpret_cast<const float*>(pblk + q * kPcmPU);
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
      // highest corner voxel a point really touches along this axis: i0 + 1 only if its fraction is non-zero (the
      // module's initial state has whole-voxel offsets, ms_deform_attn.py:67-82: seven of the eight corners then have
      // weight zero and do not widen the box -- their entries go to the spare slot below)
      const int t0 = i0[0][k] + (fl[0][k] > 0.f ? 1 : 0), t1 = i0[1][k] + (fl[1][k] > 0.f ? 1 : 0);
      const unsigned tp = __builtin_amdgcn_perm(static_cast<unsigned>(t1), static_cast<unsigned>(t0), 0x05040100u);
      const unsigned ng = __builtin_bit_cast(unsigned, -__builtin_bit_cast(s16x2, tp));
      // a skipped point is neutral (32767) in both
      mn[k] = half_min_pk16(static_cast<int>((pk & okmask) | (0x7fff7fffu & ~okmask)));
      mx[k] = half_min_pk16(static_cast<int>((ng & okmask) | (0x7fff7fffu & ~okmask)));
    }
  }

  // ---- the wave's own d-range per level, then the union of the four waves' boxes (through LDS)
  int wd_lo[kPcmLevels], wd_hi[kPcmLevels];      // corner voxels (d) the wave's points touch; lo = 32767: none
#pragma unroll
  for (int l = 0; l < kPcmLevels; ++l) {
    const int src = (l >> 1) * 32 + 31;
    const int a = __builtin_amdgcn_readlane(mn[0], src), c = __builtin_amdgcn_readlane(mx[0], src);
    wd_lo[l] = (l & 1) ? (a >> 16) : static_cast<short>(a);
    wd_hi[l] = -((l & 1) ? (c >> 16) : static_cast<int>(static_cast<short>(c)));
  }
  {
    // ubox[wave][0..5]: levels (0, 1) packed: min d, h, w, min of the negated maxima d, h, w; [6..11]: levels (2, 3);
    // [12..15]: minus the number of d-planes of the wave's own box, per level
    if (n == 31) {                    // lanes 31 / 63 hold the reductions of their halves
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        ubox[wave][kh * 6 + k] = mn[k];
        ubox[wave][kh * 6 + 3 + k] = mx[k];
      }
    }
    if (lane < kPcmLevels) {
      int td = 0;
#pragma unroll
      for (int l = 0; l < kPcmLevels; ++l) td = (lane == l) ? ((l < L && wd_lo[l] != 32767) ? wd_lo[l] - wd_hi[l] - 1 : 0) : td;
      ubox[wave][12 + lane] = td;
    }
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
  for (int i = lane; i < 32 * WP / 2; i += 64) reinterpret_cast<uint2*>(wbuf)[i] = uint2{0u, 0u};     // (the parameter block is consumed)

  __syncthreads();
  // ---- union boxes (every wave computes the same)
  PcmBox box[kPcmLevels];
  int mode[kPcmLevels];             // 0 = no valid point, 1 = staged union box, 2 = explicit (column, corner) rows per wave
  {
    int v = 0x7fff7fff;
    if (lane < 16) {
      const int a0 = ubox[0][lane], a1 = ubox[1][lane], a2 = ubox[2][lane], a3 = ubox[3][lane];
      const s16x2 m01 = __builtin_elementwise_min(__builtin_bit_cast(s16x2, a0), __builtin_bit_cast(s16x2, a1));
      const s16x2 m23 = __builtin_elementwise_min(__builtin_bit_cast(s16x2, a2), __builtin_bit_cast(s16x2, a3));
      v = __builtin_bit_cast(int, __builtin_elementwise_min(m01, m23));
    }
#pragma unroll
    for (int l = 0; l < kPcmLevels; ++l) {
      int lo3[3], hi3[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int a = __builtin_amdgcn_readlane(v, (l >> 1) * 6 + k), c = __builtin_amdgcn_readlane(v, (l >> 1) * 6 + 3 + k);
        lo3[k] = (l & 1) ? (a >> 16) : static_cast<short>(a);
        hi3[k] = -((l & 1) ? (c >> 16) : static_cast<int>(static_cast<short>(c)));
      }
      const int tdmax = -static_cast<int>(static_cast<short>(__builtin_amdgcn_readlane(v, 12 + l)));
      box[l] = PcmBox{lo3[0], lo3[1], lo3[2], hi3[0] - lo3[0] + 1, hi3[1] - lo3[1] + 1, hi3[2] - lo3[2] + 1};
      const int slab = box[l].TH * box[l].TW;
      mode[l] = (l >= L || lo3[0] == 32767) ? 0 : ((slab <= kWgbRows - 32 && tdmax * slab <= kWgbMaxK) ? 1 : 2);
    }
  }

  const unsigned rbase = static_cast<unsigned>(reinterpret_cast<size_t>((wgb_lds_void*)rbuf));
  // byte offsets of the lane's two transposing reads inside a staged row (channel tiles 0 and 1), swizzled like the rows
  const int tr_off0 = ((16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2) ^ ((((lane & 15) >> 2) & 2) << 5), tr_off1 = tr_off0 ^ 64;
  const int tr_row = 8 * kh + ((lane & 15) >> 2);                    // the lane's row inside a 16-row chunk (and + 4)
  unsigned* const wcol = wbuf + n * WP;
  const int spare = KW + (n >> 3);            // the column's spare slot (entries outside the block): the 4 slots past the K-slots, so that the 32 columns' spares sit in 32 different banks
  const int dma_piece = ((lane & 7) ^ (((lane >> 4) & 1) << 2)) * 16;     // logical 16-byte piece of its row this lane's DMA slot holds (the rows' swizzle)

  // 16-row chunks [k, k + 16 nch) of the rows at rbuf + row0 * VP against the wave's weight window
  auto mfma_block = [&](int row0, int nch) {
    for (int kc = 0; kc < nch; ++kc) {
      const unsigned* wp = wcol + kc * 16 + 8 * kh;
      const u32x4 p0 = *reinterpret_cast<const u32x4*>(wp);
      const u32x4 p1 = *reinterpret_cast<const u32x4*>(wp + 4);
      const u32x4 ahi{__builtin_amdgcn_perm(p0[1], p0[0], 0x07060302u), __builtin_amdgcn_perm(p0[3], p0[2], 0x07060302u),
                      __builtin_amdgcn_perm(p1[1], p1[0], 0x07060302u), __builtin_amdgcn_perm(p1[3], p1[2], 0x07060302u)};
      const u32x4 alo{__builtin_amdgcn_perm(p0[1], p0[0], 0x05040100u), __builtin_amdgcn_perm(p0[3], p0[2], 0x05040100u),
                      __builtin_amdgcn_perm(p1[1], p1[0], 0x05040100u), __builtin_amdgcn_perm(p1[3], p1[2], 0x05040100u)};
      const unsigned char* vrow = rbuf + (row0 + kc * 16 + tr_row) * VP;
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
  };

  // ---- box levels: rounds of (stage what fits -> wait -> barrier -> every wave walks its K range)
  int dnext[kPcmLevels] = {0, 0, 0, 0};       // first d-slab of the union box not yet staged
  bool staged_any = false;
  for (;;) {
    int pos = 0;
    bool any = false, more = false;
    int seg_d0[kPcmLevels], seg_nd[kPcmLevels], seg_base[kPcmLevels];
    static_for<0, kPcmLevels>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      seg_nd[l] = 0; seg_d0[l] = 0; seg_base[l] = 0;
      if (mode[l] != 1 || dnext[l] >= box[l].TD) return;
      const PcmBox bx = box[l];
      const int THW = bx.TH * bx.TW;
      const int room = (kWgbRows - kWgbPad - 8 - pos) / THW;
      const int nd = min(bx.TD - dnext[l], room);
      if (nd > 0) {
        const int rows = nd * THW;
        const int alloc = ((rows + 7) & ~7) + kWgbPad;                    // a multiple of 8 rows: whole DMA instructions
        seg_d0[l] = dnext[l]; seg_nd[l] = nd; seg_base[l] = pos;
        // ---- stage rows [0, alloc) of the segment.  Group g = 8 rows = one DMA instruction; wave w takes the groups
        // g = w (mod 4), 8 of them per pass: lane i computes the byte offset of row 8 (w + 4 (i >> 3)) + (i & 7) + ..., the
        // 8 lanes that fetch a row get it through ds_bpermute.  Rows past the segment or outside the level read as zeros.
        const int D = order.D[l], H = order.H[l], W = order.W[l], start = order.start[l];
        const float inv_thw = __builtin_amdgcn_rcpf(static_cast<float>(THW)), inv_tw = __builtin_amdgcn_rcpf(static_cast<float>(bx.TW));
        const int ngroups = alloc >> 3;
        for (int g0 = wave; g0 < ngroups; g0 += 4 * 8) {
          const int r = 8 * (g0 + 4 * (lane >> 3)) + (lane & 7);
          // r -> (rd, rh, rw) by float reciprocals: (r + 0.5) / n is >= 0.5 / n away from an integer, far more than
          // the float error for r < 2^12
          const int rd = static_cast<int>((static_cast<float>(r) + 0.5f) * inv_thw), rr = r - __mul24(rd, THW);
          const int rh = static_cast<int>((static_cast<float>(rr) + 0.5f) * inv_tw), rw = rr - __mul24(rh, bx.TW);
          const int d = bx.bd + dnext[l] + rd, h = bx.bh + rh, w = bx.bw + rw;
          const bool ok = r < rows && static_cast<unsigned>(d) < static_cast<unsigned>(D) && static_cast<unsigned>(h) < static_cast<unsigned>(H) &&
                          static_cast<unsigned>(w) < static_cast<unsigned>(W);
          const int grow = start + __mul24(__mul24(d, H) + h, W) + w;
          const int row_off = ok ? static_cast<int>(head_off + __umul24(static_cast<unsigned>(grow), row_bytes)) : static_cast<int>(0xffffff00u);
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const int g = g0 + 4 * t;
            if (g < ngroups) {
              const unsigned off = static_cast<unsigned>(__builtin_amdgcn_ds_bpermute((8 * t + (lane >> 3)) * 4, row_off)) + static_cast<unsigned>(dma_piece);
              wgb_dma(rsrc, off, sgpr(static_cast<int>(rbase) + (pos + 8 * g) * VP));
            }
          }
        }
        pos += alloc;
        dnext[l] += nd;
        any = true;
      }
      if (dnext[l] < bx.TD) more = true;
    });
    if (!any) break;
    staged_any = true;
    wgb_dma_wait<0>();
    __syncthreads();

    static_for<0, kPcmLevels>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      if (seg_nd[l] == 0) return;
      const PcmBox bx = box[l];
      const int THW = bx.TH * bx.TW;
      // the wave's d-range inside the segment (union-relative d-slabs [seg_d0, seg_d0 + seg_nd))
      const int d_lo = max(wd_lo[l] - bx.bd, seg_d0[l]), d_hi = min(wd_hi[l] - bx.bd + 1, seg_d0[l] + seg_nd[l]);
      if (wd_lo[l] == 32767 || d_lo >= d_hi) return;
      const int k_first = (d_lo - seg_d0[l]) * THW, k_end = (d_hi - seg_d0[l]) * THW;
      // K-slots (rows of the segment) of the lane's four entries
      int col[4];
      {
        const int pd = pdhw[l];
        const int d1 = pd & 1023, h1 = (pd >> 10) & 1023, w1 = pd >> 20;        // d0 + 1, h0 + 1, w0 + 1
        const int base = __mul24(__mul24(bx.bd + seg_d0[l] + 1, bx.TH) + (bx.bh + 1), bx.TW) + (bx.bw + 1);
        const int c0 = pd == kPcmSkip ? -1000000 : __mul24(__mul24(d1 + kh, bx.TH) + h1, bx.TW) + w1 - base;
        col[0] = c0; col[1] = c0 + 1; col[2] = c0 + bx.TW; col[3] = c0 + bx.TW + 1;
      }
      for (int k = k_first & ~3; k < k_end; k += KB) {
        int wad[4];
        bool hit = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned t = static_cast<unsigned>(col[c] - k);
          // a zero weight (a corner whose fraction is zero, a skipped point) may lie outside the box: never into a real slot
          const bool in = t < static_cast<unsigned>(KW) && wq[l][c] != 0u;
          wad[c] = in ? static_cast<int>(t) : spare;
          hit = hit || in;
        }
        if (__builtin_amdgcn_ballot_w64(hit) == 0ull) continue;          // none of the wave's entries in these 32 rows
#pragma unroll
        for (int c = 0; c < 4; ++c) wcol[wad[c]] = wq[l][c];
        mfma_block(seg_base[l] + k, (min(KB, k_end - k) + 15) >> 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) wcol[wad[c]] = 0u;
      }
    });
    if (!more) break;
    __syncthreads();                 // everyone is done with the staged rows: the next round overwrites them
  }

  // ---- explicit levels: row r of the level's 256 = (column r >> 3, corner r & 7); 32 rows per step in the wave's own
  // quarter of the row buffer, the next step's rows in flight while this one is multiplied
  if (mode[0] == 2 || mode[1] == 2 || mode[2] == 2 || mode[3] == 2) {
    if (staged_any) __syncthreads();
    const int my_row = wave * (kWgbRows / 4 & ~3);
    static_for<0, kPcmLevels>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      if (mode[l] != 2) return;
      const int D = order.D[l], H = order.H[l], W = order.W[l], start = order.start[l];
      auto issue = [&](int step) {          // rows [32 step, 32 step + 32): columns 4 step .. 4 step + 3
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int pd = __builtin_amdgcn_readlane(pdhw[l], 4 * step + i);
          const int cr = lane >> 3;           // corner dd * 4 + dh * 2 + dw
          const int d = (pd & 1023) - 1 + ((cr >> 2) & 1), h = ((pd >> 10) & 1023) - 1 + ((cr >> 1) & 1), w = ((pd >> 20) & 1023) - 1 + (cr & 1);
          const bool ok = pd != kPcmSkip && static_cast<unsigned>(d) < static_cast<unsigned>(D) && static_cast<unsigned>(h) < static_cast<unsigned>(H) &&
                          static_cast<unsigned>(w) < static_cast<unsigned>(W);
          const int grow = start + __mul24(__mul24(d, H) + h, W) + w;
          const unsigned off = ok ? head_off + __umul24(static_cast<unsigned>(grow), row_bytes) + static_cast<unsigned>(dma_piece) : 0xffffff00u;
          wgb_dma(rsrc, off, sgpr(static_cast<int>(rbase) + (my_row + (step & 1) * 32 + 8 * i) * VP));
        }
      };
      issue(0);
      for (int step = 0; step < 8; ++step) {
        if (step + 1 < 8) {
          issue(step + 1);
          wgb_dma_wait<4>();
        } else {
          wgb_dma_wait<0>();
        }
        int wad[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned t = static_cast<unsigned>(n * 8 + kh * 4 + c - step * 32);
          wad[c] = (t < static_cast<unsigned>(KW) && wq[l][c] != 0u) ? static_cast<int>(t) : spare;
          wcol[wad[c]] = wq[l][c];
        }
        mfma_block(my_row + (step & 1) * 32, 2);
#pragma unroll
        for (int c = 0; c < 4; ++c) wcol[wad[c]] = 0u;
        // the rows of this half are consumed before the DMA of step + 2 is issued: LDS reads of a wave complete in order
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    });
  }

  // ---- D[(query, point)][channel]: register r of a lane is column (r & 3) + 8 (r >> 2) + 4 kh = query 2 (r >> 2) + kh,
  // point r & 3, channel lane & 31 (+ 32 for the second tile).  Sum the points, rows leave through LDS.
  {
    unsigned short* ob = reinterpret_cast<unsigned short*>(wbuf);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float v0 = (acc0[4 * g] + acc0[4 * g + 1]) + (acc0[4 * g + 2] + acc0[4 * g + 3]);
      const float v1 = (acc1[4 * g] + acc1[4 * g + 1]) + (acc1[4 * g + 2] + acc1[4 * g + 3]);
      const unsigned pk = PcmW<VT>::pack2(v0, v1);
      ob[(2 * g + kh) * C + n] = static_cast<unsigned short>(pk);
      ob[(2 * g + kh) * C + 32 + n] = static_cast<unsigned short>(pk >> 16);
    }
    const int sq = row_of(lane >> 3);
    const u32x4 line = *reinterpret_cast<const u32x4*>(reinterpret_cast<unsigned char*>(wbuf) + (lane >> 3) * 128 + (lane & 7) * 16);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(out, 0, static_cast<int>(value_bytes), 0x00020000);
    const unsigned ooff = sq >= 0 ? (__umul24(static_cast<unsigned>(sq), static_cast<unsigned>(M)) + m) * (C * sizeof(VT)) + (lane & 7) * 16u : 0xfffffff0u;
    __builtin_amdgcn_raw_buffer_store_b128(line, ors, ooff, 0, 0);
  }
}

}  // namespace transoar
