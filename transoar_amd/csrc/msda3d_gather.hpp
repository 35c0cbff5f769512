// Query-stationary gather kernels of MSDeformAttn-3D for gfx950:
//   msda3d_fwd_vec        out = sum_{l,p} attn * trilinear(value)
//   msda3d_bwd_query_vec  grad_sampling_loc, grad_attn_weight
// (grad_value is produced by the voxel-stationary kernels in
// msda3d_scatter.hpp, without atomics.)
//
// One wave64 owns one (batch, query, head) item.
//   * geometry: lane jj of the wave does ALL the per-point arithmetic of point
//     jj of the current 16-point chunk (pixel coordinates, floor, fractions,
//     base row, corner validity bits) once; the results reach the other lanes
//     through v_readlane (SGPR broadcasts), so the per-lane work per point is
//     a handful of multiplies.
//   * data: lanes are laid out corner-group x channel-vector.  LPV = C*elt/16
//     lanes cover one voxel's head slice with 16-byte loads, so one
//     buffer_load_dwordx4 fetches 64/LPV of the 8 trilinear corners as whole
//     128/256/512-byte rows.  Loads are branch-free: an invalid corner gets an
//     out-of-range buffer offset and the hardware returns zeros, so a batch of
//     4 points issues its 4..32 loads back to back before the first FMA.
//   * reduction: L*P points and the corners accumulate in registers; one
//     xor-shuffle tree over the corner groups finishes the item.
#pragma once
#include "msda3d_common.hpp"

namespace transoar {

constexpr int kBatch = 4;  // points whose loads are issued together

// Per-point geometry, computed by lane jj for point jj of the chunk.
template <typename A> struct PointGeo {
  A ld, lh, lw, a;   // fractions along d,h,w ; attention weight
  int base_row;      // row of corner (d0,h0,w0) inside value[b] (may be < 0)
  int ok;            // bits 0-1: d lo/hi in range, 2-3: h, 4-5: w ; 0 if skipped
  int W, HW;         // row strides of the point's level
  int cell;          // padded-grid cell of (d0,h0,w0) within a (batch, head) slab, -1 if skipped
};

template <typename LT, typename A, bool WITH_CELL = false>
__device__ __forceinline__ PointGeo<A> point_geometry(const LT* __restrict__ loc_i,
                                                      const LT* __restrict__ attn_i,
                                                      const int64_t* __restrict__ shapes,
                                                      const int64_t* __restrict__ lsi, int j,
                                                      bool active, int L, int P) {
  PointGeo<A> g;
  g.ld = g.lh = g.lw = g.a = A(0);
  g.base_row = 0;
  g.ok = 0;
  g.W = g.HW = 0;
  g.cell = -1;
  if (active) {
    int l = 0;
    for (int t = 1; t < L; ++t) l += (j >= t * P) ? 1 : 0;
    const int D = static_cast<int>(shapes[3 * l]), H = static_cast<int>(shapes[3 * l + 1]),
              W = static_cast<int>(shapes[3 * l + 2]);
    const int start = static_cast<int>(lsi[l]);
    const A x = static_cast<A>(Elem<LT>::ld(loc_i + 3 * j));
    const A y = static_cast<A>(Elem<LT>::ld(loc_i + 3 * j + 1));
    const A z = static_cast<A>(Elem<LT>::ld(loc_i + 3 * j + 2));
    g.a = static_cast<A>(Elem<LT>::ld(attn_i + j));
    const A w_im = pixel_coord(x, W), h_im = pixel_coord(y, H), d_im = pixel_coord(z, D);
    g.W = W;
    g.HW = H * W;
    if (d_im > A(-1) && h_im > A(-1) && w_im > A(-1) && d_im < D && h_im < H && w_im < W) {
      const A fd = floor(d_im), fh = floor(h_im), fw = floor(w_im);
      const int d0 = static_cast<int>(fd), h0 = static_cast<int>(fh), w0 = static_cast<int>(fw);
      g.ld = d_im - fd;
      g.lh = h_im - fh;
      g.lw = w_im - fw;
      g.base_row = start + (d0 * H + h0) * W + w0;
      g.ok = (d0 >= 0 ? 1 : 0) | (d0 + 1 < D ? 2 : 0) | (h0 >= 0 ? 4 : 0) | (h0 + 1 < H ? 8 : 0) |
             (w0 >= 0 ? 16 : 0) | (w0 + 1 < W ? 32 : 0);
      if (WITH_CELL) {
        // same bin as msda3d_scatter.hpp:point_bin -- padded grid (D+1)(H+1)(W+1) per level
        int cell_start = 0;
        for (int t = 0; t < l; ++t)
          cell_start += (static_cast<int>(shapes[3 * t]) + 1) * (static_cast<int>(shapes[3 * t + 1]) + 1) *
                        (static_cast<int>(shapes[3 * t + 2]) + 1);
        g.cell = cell_start + ((d0 + 1) * (H + 1) + (h0 + 1)) * (W + 1) + (w0 + 1);
      }
    }
  }
  return g;
}

// Lane-constant description of the trilinear corner a lane fetches with its
// i-th load of a point: corner bits (dd,dh,dw), the axis weight as c + s*frac
// ((1,-1) for the low corner, (0,+1) for the high one; s is also the sign of
// d(weight)/d(frac)), row-offset masks and the validity bits it needs.
template <typename A> struct CornerConst {
  A cd, sd, ch, sh, cw, sw;
  int mask_d, mask_h, dw, need;
};
template <typename A>
__device__ __forceinline__ CornerConst<A> corner_const(int k) {
  const int dd = (k >> 2) & 1, dh = (k >> 1) & 1, dw = k & 1;
  CornerConst<A> c;
  c.cd = dd ? A(0) : A(1); c.sd = dd ? A(1) : A(-1);
  c.ch = dh ? A(0) : A(1); c.sh = dh ? A(1) : A(-1);
  c.cw = dw ? A(0) : A(1); c.sw = dw ? A(1) : A(-1);
  c.mask_d = -dd; c.mask_h = -dh; c.dw = dw;
  c.need = (1 << dd) | (4 << dh) | (16 << dw);
  return c;
}

__device__ __forceinline__ int bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

// ---------------------------------------------------------------------------
template <typename VT, typename LT, int LOG2_LPV>
__global__ __launch_bounds__(64 * kWavesPerBlock) void msda3d_fwd_vec(
    const VT* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const LT* __restrict__ loc,
    const LT* __restrict__ attn, VT* __restrict__ out, int S, int M, int C,
    int L, int Lq, int P, unsigned value_bytes, long n_units, long n_blocks, BrickOrder order) {
  using A = typename Elem<VT>::acc;
  constexpr int VEC = Elem<VT>::VEC;
  constexpr int LPV = 1 << LOG2_LPV;   // lanes per voxel row
  constexpr int CPI = 64 / LPV;        // corners per load instruction
  constexpr int NI = 8 / CPI;          // load instructions per point
  static_assert(CPI >= 1 && CPI <= 8, "lane layout");

  const long blk = xcd_contiguous_block(blockIdx.x, n_blocks);
  if (blk < 0) return;
  const int lane = threadIdx.x & 63;
  const long unit = __builtin_amdgcn_readfirstlane(
      static_cast<int>(blk * kWavesPerBlock + (threadIdx.x >> 6)));
  if (unit >= n_units) return;
  const long item = ordered_unit(order, unit, Lq, M);
  if (item < 0) return;
  const int m = static_cast<int>(item % M);
  const long b = (item / M) / Lq;
  const int cv = lane & (LPV - 1);
  const int cg = lane >> LOG2_LPV;
  const int LP = L * P;
  const unsigned row_bytes = static_cast<unsigned>(M) * C * sizeof(VT);
  const unsigned head_off = static_cast<unsigned>(((b * S * M + m) * C + cv * VEC) * sizeof(VT));
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<VT*>(value), 0, static_cast<int>(value_bytes), 0x00020000);

  CornerConst<A> corner[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) corner[i] = corner_const<A>(i * CPI + cg);
  const LT* loc_i = loc + item * LP * 3;
  const LT* attn_i = attn + item * LP;

  A acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = A(0);

  for (int j0 = 0; j0 < LP; j0 += kChunk) {
    const int nj = min(kChunk, LP - j0);
    const PointGeo<A> g = point_geometry<LT, A>(loc_i, attn_i, shapes, lsi, j0 + lane, lane < nj, L, P);
    for (int t0 = 0; t0 < nj; t0 += kBatch) {
      u32x4 raw[kBatch][NI];
      A wt[kBatch][NI];
#pragma unroll
      for (int t = 0; t < kBatch; ++t) {
        const int jj = t0 + t;   // lanes >= nj hold ok = 0, a = 0: harmless
        const A ld = bcast(g.ld, jj), lh = bcast(g.lh, jj), lw = bcast(g.lw, jj), a = bcast(g.a, jj);
        const int base = bcast(g.base_row, jj), ok = bcast(g.ok, jj);
        const int W = bcast(g.W, jj), HW = bcast(g.HW, jj);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const CornerConst<A>& cc = corner[i];
          const bool valid = (ok & cc.need) == cc.need;
          const int r = base + (HW & cc.mask_d) + (W & cc.mask_h) + cc.dw;
          const unsigned voff = valid ? head_off + __umul24(static_cast<unsigned>(r), row_bytes) : 0xfffffff0u;
          raw[t][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
          // (1-x) or x per axis as c + s*x with lane constants
          wt[t][i] = (cc.cd + cc.sd * ld) * (cc.ch + cc.sh * lh) * ((cc.cw + cc.sw * lw) * a);
        }
      }
#pragma unroll
      for (int t = 0; t < kBatch; ++t) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          A v[VEC];
          Elem<VT>::unpack(raw[t][i], v);
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] += wt[t][i] * v[e];
        }
      }
    }
  }

#pragma unroll
  for (int mask = LPV; mask < 64; mask <<= 1) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] += xor_lanes(acc[e], mask);
  }
  if (cg == 0)
    *reinterpret_cast<u32x4*>(out + item * C + cv * VEC) = Elem<VT>::pack(acc);
}

// ---------------------------------------------------------------------------
// grad_sampling_loc and grad_attn_weight.  Per point and lane four partial
// sums {attn, x, y, z}; 16 points x 4 = 64 partials are finished with one
// 63-step wave sum-transpose, after which lane r owns output r: the item's
// 16 grad_attn and 48 grad_loc values leave as two contiguous bursts.
// ---------------------------------------------------------------------------
template <typename VT, typename LT, int LOG2_LPV>
__global__ __launch_bounds__(64 * kWavesPerBlock) void msda3d_bwd_query_vec(
    const VT* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const LT* __restrict__ loc,
    const LT* __restrict__ attn, const VT* __restrict__ grad_out,
    LT* __restrict__ grad_loc, LT* __restrict__ grad_attn, int* __restrict__ bin_count,
    int* __restrict__ bin_rank, int cells_per_slab, int S, int M, int C,
    int L, int Lq, int P, unsigned value_bytes, long n_units, long n_blocks, BrickOrder order) {
  using A = typename Elem<VT>::acc;
  constexpr int VEC = Elem<VT>::VEC;
  constexpr int LPV = 1 << LOG2_LPV;
  constexpr int CPI = 64 / LPV;
  constexpr int NI = 8 / CPI;

  const long blk = xcd_contiguous_block(blockIdx.x, n_blocks);
  if (blk < 0) return;
  const int lane = threadIdx.x & 63;
  const long unit = __builtin_amdgcn_readfirstlane(
      static_cast<int>(blk * kWavesPerBlock + (threadIdx.x >> 6)));
  if (unit >= n_units) return;
  const long item = ordered_unit(order, unit, Lq, M);
  if (item < 0) return;
  const int m = static_cast<int>(item % M);
  const long b = (item / M) / Lq;
  const int cv = lane & (LPV - 1);
  const int cg = lane >> LOG2_LPV;
  const int LP = L * P;
  const unsigned row_bytes = static_cast<unsigned>(M) * C * sizeof(VT);
  const unsigned head_off = static_cast<unsigned>(((b * S * M + m) * C + cv * VEC) * sizeof(VT));
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<VT*>(value), 0, static_cast<int>(value_bytes), 0x00020000);

  CornerConst<A> corner[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) corner[i] = corner_const<A>(i * CPI + cg);
  const LT* loc_i = loc + item * LP * 3;
  const LT* attn_i = attn + item * LP;

  A go[VEC];
  Elem<VT>::unpack(*reinterpret_cast<const u32x4*>(grad_out + item * C + cv * VEC), go);

  for (int j0 = 0; j0 < LP; j0 += kChunk) {
    const int nj = min(kChunk, LP - j0);
    const PointGeo<A> g = point_geometry<LT, A, true>(loc_i, attn_i, shapes, lsi, j0 + lane, lane < nj, L, P);
    // first pass of the grad_value counting sort, folded in here: this lane already holds the
    // point's cell; the returned atomic's latency hides behind the gathers below
    if (bin_count != nullptr && lane < nj) {
      const int bin = static_cast<int>(b * M + m) * cells_per_slab + g.cell;
      bin_rank[item * LP + j0 + lane] = g.cell < 0 ? -1 : atomicAdd(bin_count + bin, 1);
    }
    A part[4 * kChunk];
#pragma unroll
    for (int i = 0; i < 4 * kChunk; ++i) part[i] = A(0);

#pragma unroll
    for (int t0 = 0; t0 < kChunk; t0 += kBatch) {
      if (t0 < nj) {
        u32x4 raw[kBatch][NI];
#pragma unroll
        for (int t = 0; t < kBatch; ++t) {
          const int jj = t0 + t;
          const int base = bcast(g.base_row, jj), ok = bcast(g.ok, jj);
          const int W = bcast(g.W, jj), HW = bcast(g.HW, jj);
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const CornerConst<A>& cc = corner[i];
            const bool valid = (ok & cc.need) == cc.need;
            const int r = base + (HW & cc.mask_d) + (W & cc.mask_h) + cc.dw;
            const unsigned voff = valid ? head_off + __umul24(static_cast<unsigned>(r), row_bytes) : 0xfffffff0u;
            raw[t][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
          }
        }
#pragma unroll
        for (int t = 0; t < kBatch; ++t) {
          const int jj = t0 + t;
          const A ld = bcast(g.ld, jj), lh = bcast(g.lh, jj), lw = bcast(g.lw, jj);
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const CornerConst<A>& cc = corner[i];
            A v[VEC];
            Elem<VT>::unpack(raw[t][i], v);
            A dot = A(0);
#pragma unroll
            for (int e = 0; e < VEC; ++e) dot += go[e] * v[e];
            const A wd = cc.cd + cc.sd * ld, wh = cc.ch + cc.sh * lh, ww = cc.cw + cc.sw * lw;
            // d(corner weight)/d(axis) = +-(product of the other two axis weights)
            part[4 * jj + 0] += (wd * wh * ww) * dot;
            part[4 * jj + 1] += (cc.sw * dot) * (wd * wh);
            part[4 * jj + 2] += (cc.sh * dot) * (wd * ww);
            part[4 * jj + 3] += (cc.sd * dot) * (wh * ww);
          }
        }
      }
    }

    const A tot = sum_transpose64(part, lane);
    // lane r -> point jj = r/4, component r%4 ; grad_loc = size * a * sum (.cuh:238-240)
    const int jj = lane >> 2, comp = lane & 3;
    const A a_jj = xor_free_shfl(g.a, jj);
    if (jj < nj) {
      const long j = item * LP + j0 + jj;
      if (comp == 0) {
        Elem<LT>::st(grad_attn + j, tot);
      } else {
        int l = 0;
        for (int t = 1; t < L; ++t) l += (j0 + jj >= t * P) ? 1 : 0;
        const int size = static_cast<int>(shapes[3 * l + 3 - comp]);   // x->W, y->H, z->D
        Elem<LT>::st(grad_loc + 3 * j + (comp - 1), tot * a_jj * static_cast<A>(size));
      }
    }
  }
}

}  // namespace transoar
