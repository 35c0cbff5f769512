// Forward gather of MSDeformAttn-3D on the matrix cores (gfx950).
//
// out[q, :] = sum over the sampling points of q of  a * w_corner * value[row, :]  is, for a group of
// neighbouring queries, a product  OUT^T[c, q] = V^T[c, r] . Wt[r, q]  of the level's value rows r inside
// the group's bounding box with a 32 x R weight matrix that has <= 32 non-zeros per query and level.  The
// per-corner formulation (msda3d_brick.hpp) spends 12 000 VALU instructions per wave on unpacking bf16
// rows and multiplying them one corner at a time; here the rows never leave their 16-bit storage form:
//
//   * one WAVE owns 32 queries (a 2x4x4 sub-brick of the pyramid) of one head; no workgroup barriers,
//   * per level it reduces the bounding box of the corner voxels its queries touch and walks the box in
//     K-blocks of 64 rows: the rows go global -> registers -> LDS as whole 128-byte head slices (prefetched
//     one block ahead, across levels), every lane adds its 16 corner weights (2 points x 8 corners) into a
//     32 x 64 fp32 weight block in LDS (read-add-write, one round per point: points of a query may share corners),
//   * per 16 rows: B = the weight block rows split into two bf16 terms (hi + lo, so the weights keep
//     16 mantissa bits: fp32-accurate results), A = V^T read straight out of the row-major LDS rows by the
//     transpose read ds_read_b64_tr_b16, 4 v_mfma_f32_32x32x16 (2 channel halves x hi/lo), fp32 accumulate,
//   * D comes out as [channel][query]: a lane holds 4 consecutive channels of its query per register
//     quad and stores 8 bytes at a time.
// A level whose box is larger than kMmaDenseRows rows (non-local sampling) is gathered corner by corner
// from global memory into the same accumulators.
//
// Dense work: 2 * 32 * R * 64 flops per wave and level on the MFMA pipe instead of 2 * 32 * 32 * 64 on the
// VALU; R is 130..250 for the self-attention pattern of the refine block.
#pragma once
#include "msda3d_common.hpp"

namespace transoar {

constexpr int kMmaKB = 64;              // value rows per K-block
constexpr int kMmaVP = 144;             // bytes per staged row: 128 + 16 (16-byte aligned, spreads banks)
constexpr int kMmaWRows = kMmaKB + 1;   // weight block [column][query]: 64 columns + one spare row for entries outside the block
constexpr int kMmaDenseRows = 1024;     // larger boxes take the per-corner path
constexpr int kMmaLevels = 4;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

template <typename VT> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  // w = hi + lo (+ 2^-16 relative): hi = w truncated to bf16, lo = (w - hi) truncated to bf16
  static __device__ __forceinline__ void split(const float (&w)[8], s16x8& hi, s16x8& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned u0 = __float_as_uint(w[2 * i]), u1 = __float_as_uint(w[2 * i + 1]);
      h[i] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
      const unsigned r0 = __float_as_uint(w[2 * i] - __uint_as_float(u0 & 0xffff0000u));
      const unsigned r1 = __float_as_uint(w[2 * i + 1] - __uint_as_float(u1 & 0xffff0000u));
      l[i] = __builtin_amdgcn_perm(r1, r0, 0x07060302u);
    }
    hi = __builtin_bit_cast(s16x8, u32x4{h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(s16x8, u32x4{l[0], l[1], l[2], l[3]});
  }
};
template <> struct Mma<f16_t> {
  static __device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ void split(const float (&w)[8], s16x8& hi, s16x8& lo) {
    f16x8_t h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      h[i] = static_cast<_Float16>(w[i]);
      l[i] = static_cast<_Float16>(w[i] - static_cast<float>(h[i]));
    }
    hi = __builtin_bit_cast(s16x8, h);
    lo = __builtin_bit_cast(s16x8, l);
  }
};

// ---- wave-wide minimum of two packed int16 (v_pk_min_i16), result wave-uniform: DPP inside the rows of
// 16 lanes, row_bcast15 / row_bcast31 across them, the total ends up in lane 63
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int wave_min_pk16(int v) {
#define TRANSOAR_DPP_STEP(CTRL, ROWMASK)                                                        \
  {                                                                                             \
    const int o = __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xf, false);                 \
    v = __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(s16x2, v), __builtin_bit_cast(s16x2, o))); \
  }
  TRANSOAR_DPP_STEP(0xB1, 0xf)    // quad_perm [1,0,3,2]
  TRANSOAR_DPP_STEP(0x4E, 0xf)    // quad_perm [2,3,0,1]
  TRANSOAR_DPP_STEP(0x141, 0xf)   // row_half_mirror
  TRANSOAR_DPP_STEP(0x140, 0xf)   // row_mirror: every lane holds its row's minimum
  TRANSOAR_DPP_STEP(0x142, 0xa)   // row_bcast15 into rows 1 and 3
  TRANSOAR_DPP_STEP(0x143, 0xc)   // row_bcast31 into rows 2 and 3
#undef TRANSOAR_DPP_STEP
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int pack16(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }

// compile-time loop: the per-level state below must stay in registers (constant indices only)
template <int I> struct IntC { static constexpr int value = I; };
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IntC<I>{});
    static_for<I + 1, N>(f);
  }
}

// geometry of one sampling point, kept for the whole kernel (5 registers)
struct MmaPoint {
  int dhw;            // (d0+1) | (h0+1) << 10 | (w0+1) << 20 ; 0x3fffffff marks a skipped point
  float ld, lh, lw, a;
};
struct MmaBox {       // wave-uniform: corner-voxel bounding box of the wave's points on one level
  int bd, bh, bw, TD, TH, TW;
};

template <typename VT, typename LT>
__global__ __launch_bounds__(64, 2) void msda3d_fwd_mma(
    const VT* __restrict__ value, const LT* __restrict__ loc, const LT* __restrict__ attn,
    VT* __restrict__ out, int S, int M, int L, unsigned value_bytes, long n_units, const BrickOrder* __restrict__ order_p) {
  const BrickOrder& order = *order_p;      // device-resident launch constants (msda3d.hip: device_const)
  constexpr int P = 4, C = 64, KB = kMmaKB, VP = kMmaVP, WR = kMmaWRows;
  __shared__ __attribute__((aligned(16))) unsigned char vbuf[KB * VP];
  __shared__ __attribute__((aligned(16))) float wbuf[WR * 32];      // [column][query]: query-fastest, bank = query

  const long u = xcd_contiguous_block(blockIdx.x, n_units);
  if (u < 0) return;
  const int lane = threadIdx.x;
  const int j = lane & 31, kg = lane >> 5;
  const int sb = static_cast<int>(u & 3);
  const long t1 = u >> 2;
  const int m = static_cast<int>(t1 % M);
  const long t2 = t1 / M;
  const int bricks = order.pad_start[order.L] >> 7;
  const int brick = bricks - 1 - static_cast<int>(t2 % bricks);      // coarse levels first: their boxes are the big ones
  const long b = t2 / bricks;
  const int slot = (2 * (sb >> 1) + (j >> 4)) * 32 + ((j >> 2) & 3) * 8 + 4 * (sb & 1) + (j & 3);
  const int s = brick_slot_to_row(order, brick * kBrickSlots + slot);
  const bool live = s >= 0;
  const long item = live ? (b * S + s) * M + m : 0;
  const int LP = L * P;
  const unsigned row_bytes = static_cast<unsigned>(M) * C * sizeof(VT);
  const unsigned head_off = static_cast<unsigned>((b * S * M + m) * C * sizeof(VT));
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<VT*>(value), 0, static_cast<int>(value_bytes), 0x00020000);

  // ---- geometry of this lane's two points (2*kg, 2*kg+1) on every level, boxes per level
  MmaPoint pt[kMmaLevels][2];
  MmaBox box[kMmaLevels];
  static_for<0, kMmaLevels>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    box[l] = MmaBox{0, 0, 0, 0, 0, 0};
    if (l >= L) return;
    const int D = order.D[l], H = order.H[l], W = order.W[l];
    int lo_d = 32767, lo_h = 32767, lo_w = 32767, hi_d = -1, hi_h = -1, hi_w = -1;
    float lx[2] = {0.f, 0.f}, ly[2] = {0.f, 0.f}, lz[2] = {0.f, 0.f}, la[2] = {0.f, 0.f};
    if (live) {
      const long jx = item * LP + l * P + 2 * kg;          // even: the two points are 24 + 8 contiguous bytes
      if constexpr (sizeof(LT) == 4) {
        const float2 q0 = *reinterpret_cast<const float2*>(loc + 3 * jx), q1 = *reinterpret_cast<const float2*>(loc + 3 * jx + 2),
                     q2 = *reinterpret_cast<const float2*>(loc + 3 * jx + 4), qa = *reinterpret_cast<const float2*>(attn + jx);
        lx[0] = q0.x; ly[0] = q0.y; lz[0] = q1.x; lx[1] = q1.y; ly[1] = q2.x; lz[1] = q2.y;
        la[0] = qa.x; la[1] = qa.y;
      } else {
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
          lx[pi] = static_cast<float>(Elem<LT>::ld(loc + 3 * (jx + pi)));
          ly[pi] = static_cast<float>(Elem<LT>::ld(loc + 3 * (jx + pi) + 1));
          lz[pi] = static_cast<float>(Elem<LT>::ld(loc + 3 * (jx + pi) + 2));
          la[pi] = static_cast<float>(Elem<LT>::ld(attn + jx + pi));
        }
      }
    }
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      MmaPoint g{0x3fffffff, 0.f, 0.f, 0.f, 0.f};
      const float w_im = pixel_coord(lx[pi], W), h_im = pixel_coord(ly[pi], H), d_im = pixel_coord(lz[pi], D);
      if (live && d_im > -1.f && h_im > -1.f && w_im > -1.f && d_im < D && h_im < H && w_im < W) {
        const float fd = floorf(d_im), fh = floorf(h_im), fw = floorf(w_im);
        const int d0 = static_cast<int>(fd), h0 = static_cast<int>(fh), w0 = static_cast<int>(fw);
        g.dhw = (d0 + 1) | ((h0 + 1) << 10) | ((w0 + 1) << 20);
        g.ld = d_im - fd; g.lh = h_im - fh; g.lw = w_im - fw; g.a = la[pi];
        lo_d = min(lo_d, max(d0, 0)); hi_d = max(hi_d, min(d0 + 1, D - 1));
        lo_h = min(lo_h, max(h0, 0)); hi_h = max(hi_h, min(h0 + 1, H - 1));
        lo_w = min(lo_w, max(w0, 0)); hi_w = max(hi_w, min(w0 + 1, W - 1));
      }
      pt[l][pi] = g;
    }
    // six wave-wide extrema as three packed 16-bit minima (maxima negated)
    const int r0 = wave_min_pk16(pack16(lo_d, lo_h)), r1 = wave_min_pk16(pack16(lo_w, -hi_d)), r2 = wave_min_pk16(pack16(-hi_h, -hi_w));
    hi_d = -(r1 >> 16);
    if (hi_d < 0) return;                         // no valid point of this wave on the level
    lo_d = static_cast<short>(r0); lo_h = r0 >> 16; lo_w = static_cast<short>(r1);
    hi_h = -static_cast<int>(static_cast<short>(r2)); hi_w = -(r2 >> 16);
    box[l] = MmaBox{lo_d, lo_h, lo_w, hi_d - lo_d + 1, hi_h - lo_h + 1, hi_w - lo_w + 1};
  });

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  for (int i = lane; i < WR * 32 / 4; i += 64) reinterpret_cast<float4*>(wbuf)[i] = float4{0.f, 0.f, 0.f, 0.f};
  for (int i = lane; i < KB * VP / 16; i += 64) reinterpret_cast<float4*>(vbuf)[i] = float4{0.f, 0.f, 0.f, 0.f};

  // rows [kb*KB, kb*KB + KB) of level l's box -> 8 x 16 bytes per lane (rows past the box: zeros).  Lane i
  // computes the byte offset of row kb*KB + i once; the 8 lanes that fetch a row get it through ds_bpermute.
  const int st_row = lane >> 3, st_vec = lane & 7;
  auto load_block = [&](auto lc, int kb, u32x4 (&pre)[8]) {
    constexpr int l = decltype(lc)::value;
    const MmaBox bx = box[l];
    const int H = order.H[l], W = order.W[l], start = order.start[l];
    const int THW = bx.TH * bx.TW, R = bx.TD * THW;
    // r -> (rd, rh, rw) by float reciprocals: (r + 0.5) / n is >= 0.5 / n away from an integer, far more than
    // the float error for r < 2^12
    const float inv_thw = __builtin_amdgcn_rcpf(static_cast<float>(THW)), inv_tw = __builtin_amdgcn_rcpf(static_cast<float>(bx.TW));   // 1 ulp
    const int r = kb * KB + lane;
    const int rd = static_cast<int>((static_cast<float>(r) + 0.5f) * inv_thw), rr = r - __mul24(rd, THW);
    const int rh = static_cast<int>((static_cast<float>(rr) + 0.5f) * inv_tw), rw = rr - __mul24(rh, bx.TW);
    const int grow = start + __mul24(__mul24(bx.bd + rd, H) + (bx.bh + rh), W) + (bx.bw + rw);
    const unsigned past = static_cast<unsigned>((R - 1 - r) >> 31);             // all ones for rows past the box
    const int row_off = static_cast<int>((head_off + __umul24(static_cast<unsigned>(grow), row_bytes)) | (past & 0xfffffff0u));
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const unsigned off = static_cast<unsigned>(__builtin_amdgcn_ds_bpermute((it * 8 + st_row) * 4, row_off)) + st_vec * 16u;
      pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    }
  };

  u32x4 pre[8];
  bool have_pre = false;
  static_for<0, kMmaLevels>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    if (l >= L || box[l].TD == 0) return;
    const MmaBox bx = box[l];
    const int D = order.D[l], H = order.H[l], W = order.W[l];
    const int THW = bx.TH * bx.TW, R = bx.TD * THW;

    // ---- weight-matrix entries of this lane: its 2 points x 8 corners.  ecol = word offset of the entry in
    // a weight block that would hold the whole box, column-major ([column][query]): column * 32 + j; a
    // corner outside the level gets a negative one.  ewt = a * trilinear weight.
    int ecol[16];
    float ewt[16];
    int near = 0;          // do the partner lane's points (same query, points +2) touch a corner of ours?
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const MmaPoint g = pt[l][pi];
      const int d0 = (g.dhw & 1023) - 1, h0 = ((g.dhw >> 10) & 1023) - 1, w0 = ((g.dhw >> 20) & 1023) - 1;
      const bool okp = g.dhw != 0x3fffffff;
      const int cb = (__mul24(__mul24(d0 - bx.bd, bx.TH) + (h0 - bx.bh), bx.TW) + (w0 - bx.bw)) * 32 + j;
      const bool vd[2] = {okp && static_cast<unsigned>(d0) < static_cast<unsigned>(D), okp && static_cast<unsigned>(d0 + 1) < static_cast<unsigned>(D)};
      const bool vh[2] = {static_cast<unsigned>(h0) < static_cast<unsigned>(H), static_cast<unsigned>(h0 + 1) < static_cast<unsigned>(H)};
      const bool vw[2] = {static_cast<unsigned>(w0) < static_cast<unsigned>(W), static_cast<unsigned>(w0 + 1) < static_cast<unsigned>(W)};
      const float wd[2] = {g.a * (1.f - g.ld), g.a * g.ld}, wh[2] = {1.f - g.lh, g.lh}, ww[2] = {1.f - g.lw, g.lw};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int dd = c >> 1, dh = c & 1;
        const float f = wd[dd] * wh[dh];
        const int col = cb + (dd ? THW * 32 : 0) + (dh ? bx.TW * 32 : 0);
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
          ecol[pi * 8 + 2 * c + dw] = (vd[dd] && vh[dh] && vw[dw]) ? col + 32 * dw : -64;
          ewt[pi * 8 + 2 * c + dw] = f * ww[dw];
        }
      }
      // point pi of lane ^ 32 is added in the same round as ours: safe only if no corner is shared
      const int odhw = __builtin_amdgcn_ds_bpermute((lane ^ 32) * 4, g.dhw);
      const int od = (odhw & 1023) - 1 - d0, oh = ((odhw >> 10) & 1023) - 1 - h0, ow = ((odhw >> 20) & 1023) - 1 - w0;
      near |= (okp && odhw != 0x3fffffff && abs(od) <= 1 && abs(oh) <= 1 && abs(ow) <= 1) ? 1 : 0;
    }
    const bool shared = __any(near);     // wave-uniform: the two halves of the wave must take turns

    if (R > kMmaDenseRows) {
      // ---- non-local level: corner by corner from global memory.  Both lanes of a query walk all 4 points
      // (the geometry of the other lane's points comes through a lane swap), each on its contiguous half of the
      // 64 channels: four 16-byte loads per corner.  The half-rows are redistributed to the D layout (a lane
      // holds channels tile*32 + 8*bq + 4*kg + (0..3)) with 16 swaps at the end of the level.
      float tacc[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) tacc[i] = 0.f;
#pragma unroll 1
      for (int p4 = 0; p4 < 4; ++p4) {
        MmaPoint g = (p4 & 1) ? pt[l][1] : pt[l][0];
        MmaPoint o;
        o.dhw = __shfl_xor(g.dhw, 32, 64);
        o.ld = __shfl_xor(g.ld, 32, 64); o.lh = __shfl_xor(g.lh, 32, 64);
        o.lw = __shfl_xor(g.lw, 32, 64); o.a = __shfl_xor(g.a, 32, 64);
        if ((p4 >> 1) != kg) g = o;
        if (g.dhw == 0x3fffffff) continue;
        const int d0 = (g.dhw & 1023) - 1, h0 = ((g.dhw >> 10) & 1023) - 1, w0 = ((g.dhw >> 20) & 1023) - 1;
#pragma unroll 1
        for (int k = 0; k < 8; ++k) {
          const int dd = k >> 2, dh = (k >> 1) & 1, dw = k & 1;
          const int d = d0 + dd, h = h0 + dh, w = w0 + dw;
          if (static_cast<unsigned>(d) >= static_cast<unsigned>(D) || static_cast<unsigned>(h) >= static_cast<unsigned>(H) ||
              static_cast<unsigned>(w) >= static_cast<unsigned>(W))
            continue;
          const float wv = g.a * ((dd ? g.ld : 1.f - g.ld) * (dh ? g.lh : 1.f - g.lh)) * (dw ? g.lw : 1.f - g.lw);
          const long grow = order.start[l] + (static_cast<long>(d) * H + h) * W + w;
          const VT* src = value + ((b * S + grow) * M + m) * C + 32 * kg;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float vv[8];
            Elem<VT>::unpack(reinterpret_cast<const u32x4*>(src)[i], vv);
#pragma unroll
            for (int e = 0; e < 8; ++e) tacc[8 * i + e] += wv * vv[e];
          }
        }
      }
#pragma unroll
      for (int bq = 0; bq < 4; ++bq) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float keep = kg == 0 ? tacc[8 * bq + t] : tacc[8 * bq + 4 + t];          // stays in this lane
          const float give = kg == 0 ? tacc[8 * bq + 4 + t] : tacc[8 * bq + t];          // belongs to the other one
          const float got = __shfl_xor(give, 32, 64);
          // lane kg = 0 holds channels 0..31: its own (8 bq + t) go to tile 0, the partner's (32 + 8 bq + t) to tile 1
          // lane kg = 1 holds channels 32..63: the partner's (8 bq + 4 + t) go to tile 0, its own (32 + 8 bq + 4 + t) to tile 1
          acc0[4 * bq + t] += kg == 0 ? keep : got;
          acc1[4 * bq + t] += kg == 0 ? got : keep;
        }
      }
      have_pre = false;
      return;
    }

    const int nblk = (R + KB - 1) / KB;
    if (!have_pre) load_block(lc, 0, pre);
    for (int kb = 0; kb < nblk; ++kb) {
      const int k0 = kb * KB;
      const int nch = (min(KB, R - k0) + 15) >> 4;          // 16-row chunks of this block
      // ---- staged rows -> LDS (rows past the box arrive as zeros)
#pragma unroll
      for (int it = 0; it < 4; ++it) *reinterpret_cast<u32x4*>(vbuf + (it * 8 + st_row) * VP + st_vec * 16) = pre[it];
      if (nch > 2) {
#pragma unroll
        for (int it = 4; it < 8; ++it) *reinterpret_cast<u32x4*>(vbuf + (it * 8 + st_row) * VP + st_vec * 16) = pre[it];
      }
      // ---- prefetch the next block (of this level, or the first of the next one)
      have_pre = false;
      if (kb + 1 < nblk) {
        load_block(lc, kb + 1, pre);
        have_pre = true;
      } else if constexpr (l + 1 < kMmaLevels) {
        if (l + 1 < L && box[l + 1].TD != 0) {
          const MmaBox nb = box[l + 1];
          if (nb.TD * nb.TH * nb.TW <= kMmaDenseRows) {
            load_block(IntC<l + 1>{}, 0, pre);
            have_pre = true;
          }
        }
      }
      // ---- weight block: W[col][j] += weight.  The 8 corners of one point are distinct words; corners of
      // different points of the query may coincide: one read-add-write round per point (LDS float atomics
      // run at a lane per clock), and the two lanes of a query take turns when their points are neighbours.
      // No predication: an entry outside this block goes to the spare row (column KB); every access of the
      // wave is one word per bank.
      int wad[16];
#pragma unroll
      for (int e = 0; e < 16; ++e)
        wad[e] = static_cast<int>(min(static_cast<unsigned>(ecol[e] - k0 * 32), static_cast<unsigned>(KB * 32 + j)));
      auto add_round = [&](auto pic, int turn) {         // turn: -1 = every lane, else the half whose turn it is
        constexpr int pi = decltype(pic)::value;
        int ad[8];
        float prev[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) ad[q] = (turn < 0 || kg == turn) ? wad[pi * 8 + q] : KB * 32 + j;
#pragma unroll
        for (int q = 0; q < 8; ++q) prev[q] = wbuf[ad[q]];
#pragma unroll
        for (int q = 0; q < 8; ++q) wbuf[ad[q]] = prev[q] + ewt[pi * 8 + q];
      };
      if (!shared) {
        add_round(IntC<0>{}, -1);
        add_round(IntC<1>{}, -1);
      } else {
        add_round(IntC<0>{}, 0);
        add_round(IntC<1>{}, 0);
        add_round(IntC<0>{}, 1);
        add_round(IntC<1>{}, 1);
      }
      // ---- pairs of 16-row chunks on the matrix cores (an odd last chunk runs on zero weights)
      for (int kc = 0; kc < nch; kc += 2) {
        s16x8 bhi[2], blo[2], a0[2], a1[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float* wcol = wbuf + ((kc + h) * 16 + 8 * kg) * 32 + j;
          float wv[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) wv[t] = wcol[32 * t];
          Mma<VT>::split(wv, bhi[h], blo[h]);
          // A = V^T: lane supplies row (lane & 15) >> 2 of its group's 4-row set, 4 channels; receives its channel's column
          const unsigned char* abase = vbuf + ((kc + h) * 16 + 8 * kg + ((lane & 15) >> 2)) * VP + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
          typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
          const s16x4 a00 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(abase));
          const s16x4 a01 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(abase + 4 * VP));
          const s16x4 a10 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(abase + 64));
          const s16x4 a11 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(abase + 4 * VP + 64));
          a0[h] = __builtin_shufflevector(a00, a01, 0, 1, 2, 3, 4, 5, 6, 7);
          a1[h] = __builtin_shufflevector(a10, a11, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          acc0 = Mma<VT>::mfma(a0[h], bhi[h], acc0);
          acc1 = Mma<VT>::mfma(a1[h], bhi[h], acc1);
          acc0 = Mma<VT>::mfma(a0[h], blo[h], acc0);
          acc1 = Mma<VT>::mfma(a1[h], blo[h], acc1);
        }
      }
      // ---- clear the weight entries again
#pragma unroll
      for (int e = 0; e < 16; ++e) wbuf[wad[e]] = 0.f;
    }
  });

  if (live) {
    VT* dst = out + item * C + 4 * kg;
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
#pragma unroll
      for (int bq = 0; bq < 4; ++bq) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          v[t] = tile == 0 ? acc0[4 * bq + t] : acc1[4 * bq + t];
          v[4 + t] = 0.f;
        }
        const u32x4 pk = Elem<VT>::pack(v);
        *reinterpret_cast<uint2*>(dst + tile * 32 + 8 * bq) = uint2{pk[0], pk[1]};
      }
    }
  }
}


// The two sampling points of lane (query j, half kg) on level l and the wave's corner-voxel bounding box: shared by
// msda3d_bwd_query_mma and the counting pre-pass of its point sort (msda3d_cell_count_mma), which must agree on the
// cell of every point bit for bit.  attn may be NULL (weights 0).  Returns false when the wave has no valid point.
template <typename LT>
__device__ __forceinline__ bool mma_level_points(const BrickOrder& order, int l, bool live, long item, int LP, int kg,
                                                 const LT* __restrict__ loc, const LT* __restrict__ attn,
                                                 MmaPoint (&pt)[2], MmaBox& box, float (&a_raw)[2]) {
  // a_raw: the attention weights as stored, ALSO of skipped points (MmaPoint::a is zero there): the softmax backward of the
  // folded sampling head needs them
  constexpr int P = 4;
  box = MmaBox{0, 0, 0, 0, 0, 0};
  a_raw[0] = a_raw[1] = 0.f;
  const int D = order.D[l], H = order.H[l], W = order.W[l];
  int lo_d = 32767, lo_h = 32767, lo_w = 32767, hi_d = -1, hi_h = -1, hi_w = -1;
  float lx[2] = {0.f, 0.f}, ly[2] = {0.f, 0.f}, lz[2] = {0.f, 0.f}, la[2] = {0.f, 0.f};
  if (live) {
    const long jx = item * LP + l * P + 2 * kg;
    if constexpr (sizeof(LT) == 4) {
      const float2 q0 = *reinterpret_cast<const float2*>(loc + 3 * jx), q1 = *reinterpret_cast<const float2*>(loc + 3 * jx + 2),
                   q2 = *reinterpret_cast<const float2*>(loc + 3 * jx + 4), qa = attn ? *reinterpret_cast<const float2*>(attn + jx) : float2{0.f, 0.f};
      lx[0] = q0.x; ly[0] = q0.y; lz[0] = q1.x; lx[1] = q1.y; ly[1] = q2.x; lz[1] = q2.y;
      la[0] = qa.x; la[1] = qa.y;
    } else {
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        lx[pi] = static_cast<float>(Elem<LT>::ld(loc + 3 * (jx + pi)));
        ly[pi] = static_cast<float>(Elem<LT>::ld(loc + 3 * (jx + pi) + 1));
        lz[pi] = static_cast<float>(Elem<LT>::ld(loc + 3 * (jx + pi) + 2));
        la[pi] = attn ? static_cast<float>(Elem<LT>::ld(attn + jx + pi)) : 0.f;
      }
    }
  }
  a_raw[0] = la[0]; a_raw[1] = la[1];
#pragma unroll
  for (int pi = 0; pi < 2; ++pi) {
    MmaPoint g{0x3fffffff, 0.f, 0.f, 0.f, 0.f};
    const float w_im = pixel_coord(lx[pi], W), h_im = pixel_coord(ly[pi], H), d_im = pixel_coord(lz[pi], D);
    if (live && d_im > -1.f && h_im > -1.f && w_im > -1.f && d_im < D && h_im < H && w_im < W) {
      const float fd = floorf(d_im), fh = floorf(h_im), fw = floorf(w_im);
      const int d0 = static_cast<int>(fd), h0 = static_cast<int>(fh), w0 = static_cast<int>(fw);
      g.dhw = (d0 + 1) | ((h0 + 1) << 10) | ((w0 + 1) << 20);
      g.ld = d_im - fd; g.lh = h_im - fh; g.lw = w_im - fw; g.a = la[pi];
      lo_d = min(lo_d, max(d0, 0)); hi_d = max(hi_d, min(d0 + 1, D - 1));
      lo_h = min(lo_h, max(h0, 0)); hi_h = max(hi_h, min(h0 + 1, H - 1));
      lo_w = min(lo_w, max(w0, 0)); hi_w = max(hi_w, min(w0 + 1, W - 1));
    }
    pt[pi] = g;
  }
  const int r0 = wave_min_pk16(pack16(lo_d, lo_h)), r1 = wave_min_pk16(pack16(lo_w, -hi_d)), r2 = wave_min_pk16(pack16(-hi_h, -hi_w));
  hi_d = -(r1 >> 16);
  if (hi_d < 0) return false;
  lo_d = static_cast<short>(r0); lo_h = r0 >> 16; lo_w = static_cast<short>(r1);
  hi_h = -static_cast<int>(static_cast<short>(r2)); hi_w = -(r2 >> 16);
  box = MmaBox{lo_d, lo_h, lo_w, hi_d - lo_d + 1, hi_h - lo_h + 1, hi_w - lo_w + 1};
  return true;
}

// ---------------------------------------------------------------------------
// grad_sampling_loc / grad_attn_weight on the matrix cores, same wave = 32 queries x 1 head structure.
// Per (point, corner) the backward needs  dot = <grad_out[q, :], value[row, :]>  -- for the 32 queries of a wave
// against the rows of their box that is the product  G[r, q] = V[r, c] . GO^T[c, q]  (64 channels = 4 MFMA
// K-steps, both operands read with their channel axis contiguous: no transposes, products of 16-bit values
// exact in fp32).  G goes through an LDS block [row][query]; every lane picks its 16 corner dots out of it
// and finishes its two points at the end of the level.  Also does the second pass of the grad_value point
// sort: it writes every point's 16-byte record (PointR16) at its sorted position -- an LDS histogram over the
// wave's cell box, then one returning global atomic per touched cell on the cells' cursors (round 3 had a third
// kernel re-read loc / attn / the ranks and write the records: 1.65 GB and 0.42 ms per call).
// ---------------------------------------------------------------------------
template <typename VT, typename LT>
__global__ __launch_bounds__(64, 2) void msda3d_bwd_query_mma(
    const VT* __restrict__ value, const LT* __restrict__ loc, const LT* __restrict__ attn,
    const VT* __restrict__ grad_out, LT* __restrict__ grad_loc, LT* __restrict__ grad_attn,
    int* __restrict__ cursor, PointR16* __restrict__ recs, unsigned long long* __restrict__ det_keys, int cells_per_slab,
    int S, int M, int L, unsigned value_bytes, long n_units, const BrickOrder* __restrict__ order_p,
    unsigned short* __restrict__ grad_proj = nullptr) {
  const BrickOrder& order = *order_p;
  constexpr int P = 4, C = 64, KB = kMmaKB, VP = kMmaVP, WR = kMmaWRows;
  __shared__ __attribute__((aligned(16))) unsigned char vbuf[KB * VP];
  __shared__ __attribute__((aligned(16))) float gbuf[WR * 32];      // G block [row][query] (+ a zero spare row); the cell histogram before that

  const long u = xcd_contiguous_block(blockIdx.x, n_units);
  if (u < 0) return;
  const int lane = threadIdx.x;
  const int j = lane & 31, kg = lane >> 5;
  const int sb = static_cast<int>(u & 3);
  const long t1 = u >> 2;
  const int m = static_cast<int>(t1 % M);
  const long t2 = t1 / M;
  const int bricks = order.pad_start[order.L] >> 7;
  const int brick = bricks - 1 - static_cast<int>(t2 % bricks);
  const long b = t2 / bricks;
  const int slot = (2 * (sb >> 1) + (j >> 4)) * 32 + ((j >> 2) & 3) * 8 + 4 * (sb & 1) + (j & 3);
  const int s = brick_slot_to_row(order, brick * kBrickSlots + slot);
  const bool live = s >= 0;
  const long item = live ? (b * S + s) * M + m : 0;
  const int LP = L * P;
  const unsigned row_bytes = static_cast<unsigned>(M) * C * sizeof(VT);
  const unsigned head_off = static_cast<unsigned>((b * S * M + m) * C * sizeof(VT));
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<VT*>(value), 0, static_cast<int>(value_bytes), 0x00020000);

  // B operand for the whole kernel: grad_out[q = j][16 s + 8 kg .. + 7], s = 0..3, packed as stored
  s16x8 gof[4];
#pragma unroll
  for (int sk = 0; sk < 4; ++sk) {
    u32x4 raw{0u, 0u, 0u, 0u};
    if (live) raw = *reinterpret_cast<const u32x4*>(grad_out + item * C + 16 * sk + 8 * kg);
    gof[sk] = __builtin_bit_cast(s16x8, raw);
  }

  MmaPoint pt[kMmaLevels][2];
  MmaBox box[kMmaLevels];
  float a_in[kMmaLevels][2];          // attention weights as stored (the folded head's softmax backward)
  static_for<0, kMmaLevels>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    box[l] = MmaBox{0, 0, 0, 0, 0, 0};
    a_in[l][0] = a_in[l][1] = 0.f;
    if (l >= L) return;
    mma_level_points<LT>(order, l, live, item, LP, kg, loc, attn, pt[l], box[l], a_in[l]);
  });

  for (int i = lane; i < WR * 32 / 4; i += 64) reinterpret_cast<float4*>(gbuf)[i] = float4{0.f, 0.f, 0.f, 0.f};
  for (int i = lane; i < KB * VP / 16; i += 64) reinterpret_cast<float4*>(vbuf)[i] = float4{0.f, 0.f, 0.f, 0.f};

  const int st_row = lane >> 3, st_vec = lane & 7;
  auto load_block = [&](auto lc, int kb, u32x4 (&pre)[8]) {
    constexpr int l = decltype(lc)::value;
    const MmaBox bx = box[l];
    const int H = order.H[l], W = order.W[l], start = order.start[l];
    const int THW = bx.TH * bx.TW, R = bx.TD * THW;
    const float inv_thw = __builtin_amdgcn_rcpf(static_cast<float>(THW)), inv_tw = __builtin_amdgcn_rcpf(static_cast<float>(bx.TW));
    const int r = kb * KB + lane;
    const int rd = static_cast<int>((static_cast<float>(r) + 0.5f) * inv_thw), rr = r - __mul24(rd, THW);
    const int rh = static_cast<int>((static_cast<float>(rr) + 0.5f) * inv_tw), rw = rr - __mul24(rh, bx.TW);
    const int grow = start + __mul24(__mul24(bx.bd + rd, H) + (bx.bh + rh), W) + (bx.bw + rw);
    const unsigned past = static_cast<unsigned>((R - 1 - r) >> 31);
    const int row_off = static_cast<int>((head_off + __umul24(static_cast<unsigned>(grow), row_bytes)) | (past & 0xfffffff0u));
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const unsigned off = static_cast<unsigned>(__builtin_amdgcn_ds_bpermute((it * 8 + st_row) * 4, row_off)) + st_vec * 16u;
      pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    }
  };

  u32x4 pre[8];
  bool have_pre = false;
  int cell_start = 0;
  // The results of a lane (per level: 2 points x (3 + 1) gradients) stay in registers until the end of the
  // wave: written level by level, the four 8..24-byte pieces of a query's 64 / 192-byte rows arrive microseconds
  // apart with 20 MB of such rows in flight on the chip, and L2 evicted the half-written lines (WRITE_SIZE 2.3 x
  // the payload).
  float res_a[kMmaLevels][2], res_l[kMmaLevels][2][3];
  static_for<0, kMmaLevels>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      res_a[l][pi] = 0.f;
      res_l[l][pi][0] = 0.f; res_l[l][pi][1] = 0.f; res_l[l][pi][2] = 0.f;
    }
  });
  static_for<0, kMmaLevels>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    if (l >= L) return;
    const int D = order.D[l], H = order.H[l], W = order.W[l];
    const int level_cells = (D + 1) * (H + 1) * (W + 1);
    const int my_cell_start = cell_start;
    cell_start += level_cells;
    if (box[l].TD == 0) return;                  // no valid point of the wave: zero gradients, no ranks
    const MmaBox bx = box[l];
    const int THW = bx.TH * bx.TW, R = bx.TD * THW;

    // ---- second pass of the point sort: the sorted position of every point of the wave and its 16-byte record.
    // cursor[cell] holds the next free position of the cell's run (the counting pre-pass msda3d_cell_count_mma and the
    // scan put the run's first position there): a wave reserves the places of all its points of a cell with ONE
    // returning atomic (LDS histogram over the wave's cell box first) and writes the records there and then.
    if (det_keys != nullptr) {
      // deterministic mode: the record goes to the point's canonical slot, together with the key (cell, slot) that the
      // stable sort of the host chain orders the points by -- no cursors, no arrival order
      const long slab_cell = static_cast<long>(b * M + m) * cells_per_slab + my_cell_start;
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const MmaPoint g = pt[l][pi];
        if (g.dhw == 0x3fffffff) continue;
        const int c1d = g.dhw & 1023, c1h = (g.dhw >> 10) & 1023, c1w = (g.dhw >> 20) & 1023;
        const unsigned long long cell = static_cast<unsigned long long>(slab_cell + (c1d * (H + 1) + c1h) * (W + 1) + c1w);
        const unsigned slot = static_cast<unsigned>(item) * static_cast<unsigned>(LP) + static_cast<unsigned>(l * P + 2 * kg + pi);
        recs[slot] = make_point_r16(g.a, static_cast<int>(item), g.ld, g.lh, g.lw);
        det_keys[slot] = (cell << 32) | slot;
      }
    } else if (cursor != nullptr) {
      int* slab_cursor = cursor + static_cast<int>(b * M + m) * cells_per_slab + my_cell_start;
      int* hist = reinterpret_cast<int*>(gbuf);
      const int CH = bx.TH + 1, CW = bx.TW + 1;
      const int ncells = (bx.TD + 1) * CH * CW;
      const bool use_hist = ncells <= KB * 32;                 // the spare row above stays zero
      int rank[2] = {-1, -1}, lcell[2] = {0, 0};
      if (use_hist) {
        for (int c = lane; c < ncells; c += 64) hist[c] = 0;
      }
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const int dhw = pt[l][pi].dhw;
        if (dhw == 0x3fffffff) continue;
        const int c1d = dhw & 1023, c1h = (dhw >> 10) & 1023, c1w = (dhw >> 20) & 1023;      // d0+1, h0+1, w0+1
        if (use_hist) {
          lcell[pi] = ((c1d - bx.bd) * CH + (c1h - bx.bh)) * CW + (c1w - bx.bw);
          rank[pi] = atomicAdd(&hist[lcell[pi]], 1);
        } else {
          rank[pi] = atomicAdd(slab_cursor + (c1d * (H + 1) + c1h) * (W + 1) + c1w, 1);
        }
      }
      if (use_hist) {
        const float inv_chw = __builtin_amdgcn_rcpf(static_cast<float>(CH * CW)), inv_cw = __builtin_amdgcn_rcpf(static_cast<float>(CW));
        // 8 cells per lane at a time with all their returning atomics in flight together (one after the other,
        // each waiting for its result before the LDS write, this loop was 4 serial round trips per level)
        for (int c0 = 0; c0 < ncells; c0 += 512) {
          int base[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int c = c0 + 64 * k + lane;
            const int n = c < ncells ? hist[c] : 0;
            base[k] = -1;
            if (n > 0) {
              const int cd = static_cast<int>((static_cast<float>(c) + 0.5f) * inv_chw), cr = c - cd * (CH * CW);
              const int ch = static_cast<int>((static_cast<float>(cr) + 0.5f) * inv_cw), cw = cr - ch * CW;
              base[k] = atomicAdd(slab_cursor + ((bx.bd + cd) * (H + 1) + (bx.bh + ch)) * (W + 1) + (bx.bw + cw), n);
            }
          }
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (base[k] >= 0) hist[c0 + 64 * k + lane] = base[k];
        }
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
          if (rank[pi] >= 0) rank[pi] += hist[lcell[pi]];
        // the histogram lived in the G block: its spare row was not touched, the rest is rewritten before use
      }
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const MmaPoint g = pt[l][pi];
        if (rank[pi] >= 0) recs[rank[pi]] = make_point_r16(g.a, static_cast<int>(item), g.ld, g.lh, g.lw);
      }
    }

    // ---- this lane's 16 corner slots: word offset in a G block that would hold the whole box
    int ecol[16];
    float dots[16];
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const MmaPoint g = pt[l][pi];
      const int d0 = (g.dhw & 1023) - 1, h0 = ((g.dhw >> 10) & 1023) - 1, w0 = ((g.dhw >> 20) & 1023) - 1;
      const bool okp = g.dhw != 0x3fffffff;
      const int cb = (__mul24(__mul24(d0 - bx.bd, bx.TH) + (h0 - bx.bh), bx.TW) + (w0 - bx.bw)) * 32 + j;
      const bool vd[2] = {okp && static_cast<unsigned>(d0) < static_cast<unsigned>(D), okp && static_cast<unsigned>(d0 + 1) < static_cast<unsigned>(D)};
      const bool vh[2] = {static_cast<unsigned>(h0) < static_cast<unsigned>(H), static_cast<unsigned>(h0 + 1) < static_cast<unsigned>(H)};
      const bool vw[2] = {static_cast<unsigned>(w0) < static_cast<unsigned>(W), static_cast<unsigned>(w0 + 1) < static_cast<unsigned>(W)};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int dd = c >> 1, dh = c & 1;
        const int col = cb + (dd ? THW * 32 : 0) + (dh ? bx.TW * 32 : 0);
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
          ecol[pi * 8 + 2 * c + dw] = (vd[dd] && vh[dh] && vw[dw]) ? col + 32 * dw : -64;
          dots[pi * 8 + 2 * c + dw] = 0.f;
        }
      }
    }

    if (R > kMmaDenseRows) {
      // ---- non-local level: corner by corner from global memory.  Both lanes of a query work on the same
      // point (each on its 32 of the 64 channels: those of its grad_out fragments), halves meet by a lane swap.
#pragma unroll 1
      for (int p4 = 0; p4 < 4; ++p4) {
        MmaPoint g = (p4 & 1) ? pt[l][1] : pt[l][0];
        MmaPoint o;
        o.dhw = __shfl_xor(g.dhw, 32, 64);
        const bool mine = (p4 >> 1) == kg;
        const int dhw = mine ? g.dhw : o.dhw;
        float part[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) part[k] = 0.f;
        if (dhw != 0x3fffffff) {
          const int d0 = (dhw & 1023) - 1, h0 = ((dhw >> 10) & 1023) - 1, w0 = ((dhw >> 20) & 1023) - 1;
#pragma unroll 1
          for (int k = 0; k < 8; ++k) {
            const int dd = k >> 2, dh = (k >> 1) & 1, dw = k & 1;
            const int d = d0 + dd, h = h0 + dh, w = w0 + dw;
            float acc = 0.f;
            if (static_cast<unsigned>(d) < static_cast<unsigned>(D) && static_cast<unsigned>(h) < static_cast<unsigned>(H) &&
                static_cast<unsigned>(w) < static_cast<unsigned>(W)) {
              const long grow = order.start[l] + (static_cast<long>(d) * H + h) * W + w;
              const VT* src = value + ((b * S + grow) * M + m) * C + 8 * kg;
#pragma unroll
              for (int sk = 0; sk < 4; ++sk) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(src + 16 * sk);
                const u32x4 gq = __builtin_bit_cast(u32x4, gof[sk]);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = Elem<VT>::dot2(v[t], gq[t], acc);
              }
            }
            // part[k] with a runtime k would index registers dynamically: select instead
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) part[kk] = kk == k ? acc : part[kk];
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float full = part[k] + __shfl_xor(part[k], 32, 64);
          // corner order of dots[]: c = dd*2 + dh, then dw  ==  k = dd*4 + dh*2 + dw
          if (mine) {
            if (p4 & 1) dots[8 + k] = full;
            else dots[k] = full;
          }
        }
      }
      have_pre = false;
    } else {
      const int nblk = (R + KB - 1) / KB;
      if (!have_pre) load_block(lc, 0, pre);
      for (int kb = 0; kb < nblk; ++kb) {
        const int k0 = kb * KB;
        const int rows_here = min(KB, R - k0);
#pragma unroll
        for (int it = 0; it < 4; ++it) *reinterpret_cast<u32x4*>(vbuf + (it * 8 + st_row) * VP + st_vec * 16) = pre[it];
        if (rows_here > 32) {
#pragma unroll
          for (int it = 4; it < 8; ++it) *reinterpret_cast<u32x4*>(vbuf + (it * 8 + st_row) * VP + st_vec * 16) = pre[it];
        }
        have_pre = false;
        if (kb + 1 < nblk) {
          load_block(lc, kb + 1, pre);
          have_pre = true;
        } else if constexpr (l + 1 < kMmaLevels) {
          if (l + 1 < L && box[l + 1].TD != 0) {
            const MmaBox nb = box[l + 1];
            if (nb.TD * nb.TH * nb.TW <= kMmaDenseRows) {
              load_block(IntC<l + 1>{}, 0, pre);
              have_pre = true;
            }
          }
        }
        // ---- G[row, q] for the 32-row tiles of the block, through LDS [row][query]
        for (int tile = 0; tile * 32 < rows_here; ++tile) {
          f32x16 acc;
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] = 0.f;
          const unsigned char* arow = vbuf + (tile * 32 + j) * VP + 16 * kg;
#pragma unroll
          for (int sk = 0; sk < 4; ++sk) {
            const s16x8 a = __builtin_bit_cast(s16x8, *reinterpret_cast<const u32x4*>(arow + 32 * sk));
            acc = Mma<VT>::mfma(a, gof[sk], acc);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) gbuf[(tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + j] = acc[r];
        }
        // ---- every lane's corner dots (a corner outside this block reads the zero spare row)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          dots[e] += gbuf[min(static_cast<unsigned>(ecol[e] - k0 * 32), static_cast<unsigned>(KB * 32 + j))];
      }
    }

    // ---- finish the two points of this lane on this level
    if (live) {
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const MmaPoint g = pt[l][pi];
        float pa = 0.f, px = 0.f, py = 0.f, pz = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int dd = k >> 2, dh = (k >> 1) & 1, dw = k & 1;
          const float dot = dots[pi * 8 + k];
          const float wd = dd ? g.ld : 1.f - g.ld, wh = dh ? g.lh : 1.f - g.lh, ww = dw ? g.lw : 1.f - g.lw;
          pa += (wd * wh * ww) * dot;
          px += (dw ? dot : -dot) * (wd * wh);
          py += (dh ? dot : -dot) * (wd * ww);
          pz += (dd ? dot : -dot) * (wh * ww);
        }
        res_a[l][pi] = pa;
        res_l[l][pi][0] = px * g.a * static_cast<float>(W);
        res_l[l][pi][1] = py * g.a * static_cast<float>(H);
        res_l[l][pi][2] = pz * g.a * static_cast<float>(D);
      }
    }
  });

  if constexpr (sizeof(LT) == 4) {
    if (grad_proj != nullptr) {
      // ---- the sampling head's backward folded in (csrc/tokens.hip: sampling_head_bwd_p16; reference: the autograd of
      // ops/modules/ms_deform_attn.py:114-128): instead of 64 fp32 gradients per (query, head) -- 360 MB written here and
      // re-read by a second kernel -- the wave writes the gradient of the STACKED bf16 projection:
      //   offsets  bf16( bf16(grad_loc) / bf16(W | H | D) )                          columns m 48 + l 12 + p 3 + k
      //   logits   bf16( a (grad_attn - sum_{l,p} a grad_attn) )   (softmax backward)  columns 48 M + m 16 + l 4 + p
      // A lane holds points 2 kg, 2 kg + 1 of every level; its partner (lane ^ 32) the other two.  The 32 queries' 128
      // bytes each go through LDS and leave as 16-byte pieces of the projection rows.
      float part = 0.f;
      static_for<0, kMmaLevels>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        if (l >= L) return;
        part = fmaf(a_in[l][0], res_a[l][0], part);
        part = fmaf(a_in[l][1], res_a[l][1], part);
      });
      const float dot = part + __shfl_xor(part, 32, 64);
      unsigned* stage = reinterpret_cast<unsigned*>(vbuf);              // [query][32 dwords]: 24 of offsets, 8 of logits
      int* tok = reinterpret_cast<int*>(vbuf + 32 * 128);               // projection row of the query, -1 = padding
      auto bf = [](float x) -> unsigned { return static_cast<unsigned>(f32_to_bf16(x)); };
      auto rnd = [](float x) -> float { return bf16_to_f32(f32_to_bf16(x)); };
      static_for<0, kMmaLevels>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        if (l >= L) return;
        const float sz[3] = {rnd(static_cast<float>(order.W[l])), rnd(static_cast<float>(order.H[l])), rnd(static_cast<float>(order.D[l]))};
        // x / size correctly rounded without the 10-instruction IEEE division: quotient estimate + one residual step (the
        // divisor is a small integer, wave-uniform: its reciprocal is scalar work) -- msda3d_pcm.hpp's fused head does the same
        const float rc[3] = {__builtin_amdgcn_rcpf(sz[0]), __builtin_amdgcn_rcpf(sz[1]), __builtin_amdgcn_rcpf(sz[2])};
        unsigned o[6];
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float x = rnd(res_l[l][pi][k]);
            const float q0 = x * rc[k];
            o[3 * pi + k] = bf(__builtin_fmaf(__builtin_fmaf(-q0, sz[k], x), rc[k], q0));
          }
        unsigned* dst = stage + j * 32 + l * 6 + kg * 3;
        dst[0] = o[0] | (o[1] << 16);
        dst[1] = o[2] | (o[3] << 16);
        dst[2] = o[4] | (o[5] << 16);
        stage[j * 32 + 24 + l * 2 + kg] = bf(a_in[l][0] * (res_a[l][0] - dot)) | (bf(a_in[l][1] * (res_a[l][1] - dot)) << 16);
      });
      if (kg == 0) tok[j] = live ? static_cast<int>(b * S + s) : -1;
      __syncthreads();          // one wave per workgroup: orders the LDS writes above before the reads below (also for the compiler)
      const unsigned row_pieces = static_cast<unsigned>(M) * 8u;          // 16-byte pieces per projection row: 6 M of offsets, 2 M of logits
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int e = it * 64 + lane, q = e >> 3, pc = e & 7;
        const int t = tok[q];
        if (t < 0) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(vbuf + q * 128 + pc * 16);
        const unsigned piece = pc < 6 ? static_cast<unsigned>(m) * 6u + pc : static_cast<unsigned>(M) * 6u + static_cast<unsigned>(m) * 2u + (pc - 6);
        *reinterpret_cast<u32x4*>(grad_proj + (static_cast<long>(t) * row_pieces + piece) * 8) = v;
      }
      return;
    }
  }
  if (live) {
    static_for<0, kMmaLevels>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      if (l >= L) return;
      const long jx = item * LP + l * P + 2 * kg;          // even: 8-byte aligned pairs
      if constexpr (sizeof(LT) == 4) {
        *reinterpret_cast<float2*>(grad_attn + jx) = float2{res_a[l][0], res_a[l][1]};
        float2* gl = reinterpret_cast<float2*>(grad_loc + 3 * jx);
        gl[0] = float2{res_l[l][0][0], res_l[l][0][1]};
        gl[1] = float2{res_l[l][0][2], res_l[l][1][0]};
        gl[2] = float2{res_l[l][1][1], res_l[l][1][2]};
      } else {
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
          Elem<LT>::st(grad_attn + jx + pi, res_a[l][pi]);
#pragma unroll
          for (int k = 0; k < 3; ++k) Elem<LT>::st(grad_loc + 3 * (jx + pi) + k, res_l[l][pi][k]);
        }
      }
    });
  }
}

// ---------------------------------------------------------------------------
// Counting pre-pass of the point sort: the number of valid sampling points per cell, with the unit decomposition and
// the geometry (mma_level_points) of msda3d_bwd_query_mma, so that both passes see the same cell for every point.
// Reads the locations only (270 MB at the flagship size); an LDS histogram over the wave's cell box, then ONE
// non-returning global atomic per touched cell.  count = the slot of cell 0 (the caller shifts the array by one
// entry, so that the exclusive scan leaves every cell's first position in its cursor slot).
// ---------------------------------------------------------------------------
template <typename LT>
__global__ __launch_bounds__(64) void msda3d_cell_count_mma(
    const LT* __restrict__ loc, int* __restrict__ count, int cells_per_slab, int S, int M, int L, long n_units,
    const BrickOrder* __restrict__ order_p) {
  const BrickOrder& order = *order_p;
  constexpr int P = 4, KB = kMmaKB;
  __shared__ int hist[KB * 32];

  const long u = xcd_contiguous_block(blockIdx.x, n_units);
  if (u < 0) return;
  const int lane = threadIdx.x;
  const int j = lane & 31, kg = lane >> 5;
  const int sb = static_cast<int>(u & 3);
  const long t1 = u >> 2;
  const int m = static_cast<int>(t1 % M);
  const long t2 = t1 / M;
  const int bricks = order.pad_start[order.L] >> 7;
  const int brick = bricks - 1 - static_cast<int>(t2 % bricks);
  const long b = t2 / bricks;
  const int slot = (2 * (sb >> 1) + (j >> 4)) * 32 + ((j >> 2) & 3) * 8 + 4 * (sb & 1) + (j & 3);
  const int s = brick_slot_to_row(order, brick * kBrickSlots + slot);
  const bool live = s >= 0;
  const long item = live ? (b * S + s) * M + m : 0;
  const int LP = L * P;

  int cell_start = 0;
  static_for<0, kMmaLevels>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    if (l >= L) return;
    const int D = order.D[l], H = order.H[l], W = order.W[l];
    const int my_cell_start = cell_start;
    cell_start += (D + 1) * (H + 1) * (W + 1);
    MmaPoint pt[2];
    MmaBox bx;
    float a_unused[2];
    if (!mma_level_points<LT>(order, l, live, item, LP, kg, loc, static_cast<const LT*>(nullptr), pt, bx, a_unused)) return;
    int* slab_count = count + static_cast<int>(b * M + m) * cells_per_slab + my_cell_start;
    const int CH = bx.TH + 1, CW = bx.TW + 1;
    const int ncells = (bx.TD + 1) * CH * CW;
    const bool use_hist = ncells <= KB * 32;
    if (use_hist) {
      for (int c = lane; c < ncells; c += 64) hist[c] = 0;
    }
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const int dhw = pt[pi].dhw;
      if (dhw == 0x3fffffff) continue;
      const int c1d = dhw & 1023, c1h = (dhw >> 10) & 1023, c1w = (dhw >> 20) & 1023;      // d0+1, h0+1, w0+1
      if (use_hist) atomicAdd(&hist[((c1d - bx.bd) * CH + (c1h - bx.bh)) * CW + (c1w - bx.bw)], 1);
      else atomicAdd(slab_count + (c1d * (H + 1) + c1h) * (W + 1) + c1w, 1);
    }
    if (use_hist) {
      const float inv_chw = __builtin_amdgcn_rcpf(static_cast<float>(CH * CW)), inv_cw = __builtin_amdgcn_rcpf(static_cast<float>(CW));
      for (int c = lane; c < ncells; c += 64) {
        const int n = hist[c];
        if (n > 0) {
          const int cd = static_cast<int>((static_cast<float>(c) + 0.5f) * inv_chw), cr = c - cd * (CH * CW);
          const int ch = static_cast<int>((static_cast<float>(cr) + 0.5f) * inv_cw), cw = cr - ch * CW;
          atomicAdd(slab_count + ((bx.bd + cd) * (H + 1) + (bx.bh + ch)) * (W + 1) + (bx.bw + cw), n);
        }
      }
    }
  });
}

}  // namespace transoar
