// LDS-tiled forward gather of MSDeformAttn-3D (gfx950).
//
// The per-item kernel (msda3d_gather.hpp) pushes every one of the 128 corner
// rows of an output row through the vector L1 and is VALU-bound on the per-lane
// address/weight arithmetic (PMC: profiles/r01_msda_pmc.json).  Here a
// workgroup owns one 4x4x8 brick of queries (= voxels of the pyramid, the
// refine block's self-attention) for ONE head.  Per level it
//   1. computes the sampling geometry of its 128 queries and the bounding box
//      of all their corner voxels (wave shuffles + LDS min/max),
//   2. copies that box of the level -- whole head slices, coalesced 16-byte
//      loads -- into LDS once (a few hundred rows instead of 128*P*8 = 4096
//      row fetches),
//   3. lets every thread (query x channel-half) accumulate its corners out of
//      LDS with ds_read_b128: no cross-lane reduction, a quarter of the VALU
//      work per item.
// A level whose box does not fit the tile (non-local sampling patterns, e.g.
// the reference test's uniform locations) is gathered from global memory by
// the same threads -- slower, same result.  Needs the host-side level shapes
// (brick schedule) and Lq == S; everything else stays on the per-item kernel.
#pragma once
#include "msda3d_common.hpp"

namespace transoar {

constexpr int kBrickThreads = 256;
constexpr int kHistCells = 1024;              // LDS counters of the binning pass (cells of one level's box)            // 128 queries x 2 channel halves
constexpr int kTileBytes = 48 * 1024;         // LDS per workgroup -> 3 workgroups per CU

template <typename VT> struct BrickTraits {
  using A = typename Elem<VT>::acc;
  static constexpr int VEC = Elem<VT>::VEC;                  // elements per 16 bytes
};

// value row -> CPT accumulators, from LDS or global (same code path)
template <typename VT, int NV>
__device__ __forceinline__ void fma_row(const u32x4* __restrict__ src, typename Elem<VT>::acc wt,
                                        typename Elem<VT>::acc (&acc)[NV * Elem<VT>::VEC]) {
  constexpr int VEC = Elem<VT>::VEC;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    typename Elem<VT>::acc v[VEC];
    Elem<VT>::unpack(src[i], v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[i * VEC + e] += wt * v[e];
  }
}

template <typename VT, typename LT, int P, int C>
__global__ __launch_bounds__(kBrickThreads) void msda3d_fwd_brick(
    const VT* __restrict__ value, const LT* __restrict__ loc, const LT* __restrict__ attn,
    VT* __restrict__ out, int S, int M, int L, long n_wg, BrickOrder order) {
  using A = typename Elem<VT>::acc;
  constexpr int VEC = Elem<VT>::VEC;
  constexpr int CPT = C / 2;                  // channels per thread
  constexpr int NV = CPT / VEC;               // 16-byte vectors per thread per row
  constexpr int ROW_BYTES = C * sizeof(VT);
  constexpr int PITCH = ROW_BYTES + 16;       // padded LDS row: spreads rows over the banks
  constexpr int ROW_VECS = ROW_BYTES / 16;
  constexpr int TILE_ROWS = kTileBytes / PITCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char tile[];
  __shared__ int box[6];                      // min d,h,w ; max d,h,w of the corner voxels

  const long wg = xcd_contiguous_block(blockIdx.x, n_wg);
  if (wg < 0) return;
  const int tid = threadIdx.x;
  const int m = static_cast<int>(wg % M);
  const long t1 = wg / M;
  const int bricks = order.pad_start[order.L] >> 7;
  const int brick = static_cast<int>(t1 % bricks);
  const long b = t1 / bricks;
  const int q_slot = tid >> 1, half = tid & 1;
  const int s = brick_slot_to_row(order, brick * kBrickSlots + q_slot);   // per-thread (not uniform here)
  const bool live = s >= 0;
  const long item = live ? (b * S + s) * M + m : 0;
  const int LP = L * P;
  const long row_stride = static_cast<long>(M) * C;
  const VT* vhead = value + (b * S * M + m) * C + half * CPT;

  A acc[CPT];
#pragma unroll
  for (int e = 0; e < CPT; ++e) acc[e] = A(0);

  for (int l = 0; l < L; ++l) {
    const int D = order.D[l], H = order.H[l], W = order.W[l], start = order.start[l];
    // ---- geometry of this thread's P points on level l
    A ld[P], lh[P], lw[P], aw[P];
    int d0[P], h0[P], w0[P];
    bool ok[P];
    int lo_d = 1 << 30, lo_h = 1 << 30, lo_w = 1 << 30, hi_d = -1, hi_h = -1, hi_w = -1;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      ok[p] = false;
      ld[p] = lh[p] = lw[p] = aw[p] = A(0);
      d0[p] = h0[p] = w0[p] = 0;
      if (live) {
        const long j = item * LP + l * P + p;
        const A w_im = pixel_coord(static_cast<A>(Elem<LT>::ld(loc + 3 * j)), W);
        const A h_im = pixel_coord(static_cast<A>(Elem<LT>::ld(loc + 3 * j + 1)), H);
        const A d_im = pixel_coord(static_cast<A>(Elem<LT>::ld(loc + 3 * j + 2)), D);
        if (d_im > A(-1) && h_im > A(-1) && w_im > A(-1) && d_im < D && h_im < H && w_im < W) {
          const A fd = floor(d_im), fh = floor(h_im), fw = floor(w_im);
          d0[p] = static_cast<int>(fd); h0[p] = static_cast<int>(fh); w0[p] = static_cast<int>(fw);
          ld[p] = d_im - fd; lh[p] = h_im - fh; lw[p] = w_im - fw;
          aw[p] = static_cast<A>(Elem<LT>::ld(attn + item * LP + l * P + p));
          ok[p] = true;
          lo_d = min(lo_d, max(d0[p], 0)); hi_d = max(hi_d, min(d0[p] + 1, D - 1));
          lo_h = min(lo_h, max(h0[p], 0)); hi_h = max(hi_h, min(h0[p] + 1, H - 1));
          lo_w = min(lo_w, max(w0[p], 0)); hi_w = max(hi_w, min(w0[p] + 1, W - 1));
        }
      }
    }
    // ---- bounding box of the workgroup's corners
    if (tid < 3) box[tid] = 1 << 30;
    else if (tid < 6) box[tid] = -1;
    __syncthreads();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      lo_d = min(lo_d, __shfl_xor(lo_d, off, 64)); hi_d = max(hi_d, __shfl_xor(hi_d, off, 64));
      lo_h = min(lo_h, __shfl_xor(lo_h, off, 64)); hi_h = max(hi_h, __shfl_xor(hi_h, off, 64));
      lo_w = min(lo_w, __shfl_xor(lo_w, off, 64)); hi_w = max(hi_w, __shfl_xor(hi_w, off, 64));
    }
    if ((tid & 63) == 0) {
      atomicMin(&box[0], lo_d); atomicMin(&box[1], lo_h); atomicMin(&box[2], lo_w);
      atomicMax(&box[3], hi_d); atomicMax(&box[4], hi_h); atomicMax(&box[5], hi_w);
    }
    __syncthreads();
    const int bd = box[0], bh = box[1], bw = box[2];
    const int TD = box[3] - bd + 1, TH = box[4] - bh + 1, TW = box[5] - bw + 1;
    if (box[3] < 0) { __syncthreads(); continue; }          // no valid point on this level (uniform)
    const int rows = TD * TH * TW;
    const bool staged = rows <= TILE_ROWS;                   // uniform
    if (staged) {
      // ---- copy the box into LDS: whole head slices, 16 bytes per thread
      const int THW = TH * TW;
      // r -> (rd, rh, rw) with two float multiplies instead of two runtime integer divisions (r < 2^12:
      // (r + 0.5) / n is at least 0.5 / n away from an integer, far more than the float error)
      const float inv_thw = 1.0f / static_cast<float>(THW), inv_tw = 1.0f / static_cast<float>(TW);
      for (int i = tid; i < rows * ROW_VECS; i += kBrickThreads) {
        const int r = i / ROW_VECS, v = i - r * ROW_VECS;
        const int rd = static_cast<int>((static_cast<float>(r) + 0.5f) * inv_thw), rr = r - rd * THW;
        const int rh = static_cast<int>((static_cast<float>(rr) + 0.5f) * inv_tw), rw = rr - rh * TW;
        const long grow = start + (static_cast<long>(bd + rd) * H + (bh + rh)) * W + (bw + rw);
        const u32x4 x = *reinterpret_cast<const u32x4*>(value + ((b * S + grow) * M + m) * C + v * VEC);
        *reinterpret_cast<u32x4*>(tile + r * PITCH + v * 16) = x;
      }
      __syncthreads();
    }
    // ---- sample
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (!ok[p]) continue;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int dd = k >> 2, dh = (k >> 1) & 1, dw = k & 1;
        const int d = d0[p] + dd, h = h0[p] + dh, w = w0[p] + dw;
        if (static_cast<unsigned>(d) < static_cast<unsigned>(D) && static_cast<unsigned>(h) < static_cast<unsigned>(H) &&
            static_cast<unsigned>(w) < static_cast<unsigned>(W)) {
          const A wt = (dd ? ld[p] : A(1) - ld[p]) * (dh ? lh[p] : A(1) - lh[p]) * (dw ? lw[p] : A(1) - lw[p]) * aw[p];
          if (staged) {
            const int r = ((d - bd) * TH + (h - bh)) * TW + (w - bw);
            fma_row<VT, NV>(reinterpret_cast<const u32x4*>(tile + r * PITCH + half * (CPT * sizeof(VT))), wt, acc);
          } else {
            const long grow = start + (static_cast<long>(d) * H + h) * W + w;
            fma_row<VT, NV>(reinterpret_cast<const u32x4*>(vhead + grow * row_stride), wt, acc);
          }
        }
      }
    }
    __syncthreads();        // the tile is rewritten by the next level
  }

  if (live) {
    VT* dst = out + item * C + half * CPT;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      A v[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = acc[i * VEC + e];
      reinterpret_cast<u32x4*>(dst)[i] = Elem<VT>::pack(v);
    }
  }
}

// ---------------------------------------------------------------------------
// grad_sampling_loc / grad_attn_weight, same tiling.  A thread holds the
// grad_out half-row of its query; per corner it dots it with the LDS row, the
// two halves of a query meet through one lane exchange per point, and half 0
// writes the point's 4 gradients.  Also does the binning pass of the
// grad_value point sort (one int atomic per point, see msda3d_scatter.hpp).
// ---------------------------------------------------------------------------
// both operands stay in their packed 16-bit storage form: 4 two-element dot instructions per
// 16 bytes (products exact, fp32 accumulation) instead of 8 unpacks + 8 multiply-adds
template <typename VT, int NV>
__device__ __forceinline__ float dot_row(const u32x4* __restrict__ src, const u32x4 (&go)[NV]) {
  float d0 = 0.f, d1 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const u32x4 v = src[i];
    d0 = Elem<VT>::dot2(v[0], go[i][0], d0);
    d1 = Elem<VT>::dot2(v[1], go[i][1], d1);
    d0 = Elem<VT>::dot2(v[2], go[i][2], d0);
    d1 = Elem<VT>::dot2(v[3], go[i][3], d1);
  }
  return d0 + d1;
}

template <typename VT, typename LT, int P, int C>
__global__ __launch_bounds__(kBrickThreads) void msda3d_bwd_query_brick(
    const VT* __restrict__ value, const LT* __restrict__ loc, const LT* __restrict__ attn,
    const VT* __restrict__ grad_out, LT* __restrict__ grad_loc, LT* __restrict__ grad_attn,
    int* __restrict__ bin_count, int* __restrict__ bin_rank, int cells_per_slab, int S, int M, int L,
    long n_wg, BrickOrder order) {
  using A = typename Elem<VT>::acc;
  constexpr int VEC = Elem<VT>::VEC;
  constexpr int CPT = C / 2;
  constexpr int NV = CPT / VEC;
  constexpr int ROW_BYTES = C * sizeof(VT);
  constexpr int PITCH = ROW_BYTES + 16;
  constexpr int ROW_VECS = ROW_BYTES / 16;
  constexpr int TILE_ROWS = kTileBytes / PITCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char tile[];
  __shared__ int box[6];
  __shared__ int hist[kHistCells];

  const long wg = xcd_contiguous_block(blockIdx.x, n_wg);
  if (wg < 0) return;
  const int tid = threadIdx.x;
  const int m = static_cast<int>(wg % M);
  const long t1 = wg / M;
  const int bricks = order.pad_start[order.L] >> 7;
  const int brick = static_cast<int>(t1 % bricks);
  const long b = t1 / bricks;
  const int q_slot = tid >> 1, half = tid & 1;
  const int s = brick_slot_to_row(order, brick * kBrickSlots + q_slot);
  const bool live = s >= 0;
  const long item = live ? (b * S + s) * M + m : 0;
  const int LP = L * P;
  const long row_stride = static_cast<long>(M) * C;
  const VT* vhead = value + (b * S * M + m) * C + half * CPT;

  u32x4 go[NV];                 // the query's grad_out half-row, packed as stored
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    go[i] = u32x4{0u, 0u, 0u, 0u};
    if (live) go[i] = reinterpret_cast<const u32x4*>(grad_out + item * C + half * CPT)[i];
  }

  int cell_start = 0;
  for (int l = 0; l < L; ++l) {
    const int D = order.D[l], H = order.H[l], W = order.W[l], start = order.start[l];
    A ld[P], lh[P], lw[P], aw[P];
    int d0[P], h0[P], w0[P], rank[P];
    bool ok[P];
    int lo_d = 1 << 30, lo_h = 1 << 30, lo_w = 1 << 30, hi_d = -1, hi_h = -1, hi_w = -1;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      ok[p] = false;
      rank[p] = -1;
      ld[p] = lh[p] = lw[p] = aw[p] = A(0);
      d0[p] = h0[p] = w0[p] = 0;
      if (live) {
        const long j = item * LP + l * P + p;
        const A w_im = pixel_coord(static_cast<A>(Elem<LT>::ld(loc + 3 * j)), W);
        const A h_im = pixel_coord(static_cast<A>(Elem<LT>::ld(loc + 3 * j + 1)), H);
        const A d_im = pixel_coord(static_cast<A>(Elem<LT>::ld(loc + 3 * j + 2)), D);
        if (d_im > A(-1) && h_im > A(-1) && w_im > A(-1) && d_im < D && h_im < H && w_im < W) {
          const A fd = floor(d_im), fh = floor(h_im), fw = floor(w_im);
          d0[p] = static_cast<int>(fd); h0[p] = static_cast<int>(fh); w0[p] = static_cast<int>(fw);
          ld[p] = d_im - fd; lh[p] = h_im - fh; lw[p] = w_im - fw;
          aw[p] = static_cast<A>(Elem<LT>::ld(attn + j));
          ok[p] = true;
          lo_d = min(lo_d, max(d0[p], 0)); hi_d = max(hi_d, min(d0[p] + 1, D - 1));
          lo_h = min(lo_h, max(h0[p], 0)); hi_h = max(hi_h, min(h0[p] + 1, H - 1));
          lo_w = min(lo_w, max(w0[p], 0)); hi_w = max(hi_w, min(w0[p] + 1, W - 1));
        }
      }
    }
    if (tid < 3) box[tid] = 1 << 30;
    else if (tid < 6) box[tid] = -1;
#pragma unroll
    for (int i = 0; i < kHistCells / kBrickThreads; ++i) hist[tid + i * kBrickThreads] = 0;
    __syncthreads();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      lo_d = min(lo_d, __shfl_xor(lo_d, off, 64)); hi_d = max(hi_d, __shfl_xor(hi_d, off, 64));
      lo_h = min(lo_h, __shfl_xor(lo_h, off, 64)); hi_h = max(hi_h, __shfl_xor(hi_h, off, 64));
      lo_w = min(lo_w, __shfl_xor(lo_w, off, 64)); hi_w = max(hi_w, __shfl_xor(hi_w, off, 64));
    }
    if ((tid & 63) == 0) {
      atomicMin(&box[0], lo_d); atomicMin(&box[1], lo_h); atomicMin(&box[2], lo_w);
      atomicMax(&box[3], hi_d); atomicMax(&box[4], hi_h); atomicMax(&box[5], hi_w);
    }
    __syncthreads();
    const int bd = box[0], bh = box[1], bw = box[2];
    const int TD = box[3] - bd + 1, TH = box[4] - bh + 1, TW = box[5] - bw + 1;
    const bool any = box[3] >= 0;
    const int rows = TD * TH * TW;
    const bool staged = any && rows <= TILE_ROWS;

    // Binning pass of the grad_value point sort: rank of every point inside its cell.  The
    // points of this workgroup are first counted per cell in LDS (cells of the box: floor+1 on
    // each axis spans one more than the voxel box), then ONE global atomic per touched cell
    // reserves the workgroup's range -- a coarse cell receives thousands of points per slab,
    // and one returning atomic per point on such a hot counter costs as much as the rest of
    // this kernel.  The global atomics are in flight while the tile is staged.
    int* slab_count = bin_count == nullptr ? nullptr : bin_count + static_cast<int>(b * M + m) * cells_per_slab + cell_start;
    const int CH = TH + 1, CW = TW + 1;
    const int ncells = (TD + 1) * CH * CW;
    const bool use_hist = any && ncells <= kHistCells;
    int lcell[P], cell_base[kHistCells / kBrickThreads];
    if (slab_count != nullptr && half == 0) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        lcell[p] = 0;
        if (!ok[p]) continue;
        if (use_hist) {
          lcell[p] = ((d0[p] + 1 - bd) * CH + (h0[p] + 1 - bh)) * CW + (w0[p] + 1 - bw);
          rank[p] = atomicAdd(&hist[lcell[p]], 1);
        } else {
          rank[p] = atomicAdd(slab_count + ((d0[p] + 1) * (H + 1) + (h0[p] + 1)) * (W + 1) + (w0[p] + 1), 1);
        }
      }
    }
    if (slab_count != nullptr && use_hist) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < kHistCells / kBrickThreads; ++i) {
        const int c = tid + i * kBrickThreads;
        cell_base[i] = 0;
        const int n = c < ncells ? hist[c] : 0;
        if (n > 0) {
          const int cd = c / (CH * CW), cr = c - cd * (CH * CW);
          const int ch = cr / CW, cw = cr - ch * CW;
          cell_base[i] = atomicAdd(slab_count + ((bd + cd) * (H + 1) + (bh + ch)) * (W + 1) + (bw + cw), n);
        }
      }
    }
    cell_start += (D + 1) * (H + 1) * (W + 1);

    if (staged) {
      const int THW = TH * TW;
      const float inv_thw = 1.0f / static_cast<float>(THW), inv_tw = 1.0f / static_cast<float>(TW);   // see the forward
      for (int i = tid; i < rows * ROW_VECS; i += kBrickThreads) {
        const int r = i / ROW_VECS, v = i - r * ROW_VECS;
        const int rd = static_cast<int>((static_cast<float>(r) + 0.5f) * inv_thw), rr = r - rd * THW;
        const int rh = static_cast<int>((static_cast<float>(rr) + 0.5f) * inv_tw), rw = rr - rh * TW;
        const long grow = start + (static_cast<long>(bd + rd) * H + (bh + rh)) * W + (bw + rw);
        *reinterpret_cast<u32x4*>(tile + r * PITCH + v * 16) =
            *reinterpret_cast<const u32x4*>(value + ((b * S + grow) * M + m) * C + v * VEC);
      }
    }
    if (slab_count != nullptr && use_hist) {
#pragma unroll
      for (int i = 0; i < kHistCells / kBrickThreads; ++i) {
        const int c = tid + i * kBrickThreads;
        if (c < ncells) hist[c] = cell_base[i];
      }
    }
    __syncthreads();          // tile staged, cell bases published
    if (slab_count != nullptr && use_hist && half == 0) {
#pragma unroll
      for (int p = 0; p < P; ++p)
        if (ok[p]) rank[p] += hist[lcell[p]];
    }
    A res_loc[3 * P], res_attn[P];     // this level's outputs of the query: 48 + 16 contiguous bytes
#pragma unroll
    for (int p = 0; p < P; ++p) {
      A pa = A(0), px = A(0), py = A(0), pz = A(0);
      if (ok[p]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int dd = k >> 2, dh = (k >> 1) & 1, dw = k & 1;
          const int d = d0[p] + dd, h = h0[p] + dh, w = w0[p] + dw;
          if (static_cast<unsigned>(d) < static_cast<unsigned>(D) && static_cast<unsigned>(h) < static_cast<unsigned>(H) &&
              static_cast<unsigned>(w) < static_cast<unsigned>(W)) {
            A dot;
            if (staged) {
              const int r = ((d - bd) * TH + (h - bh)) * TW + (w - bw);
              dot = dot_row<VT, NV>(reinterpret_cast<const u32x4*>(tile + r * PITCH + half * (CPT * sizeof(VT))), go);
            } else {
              const long grow = start + (static_cast<long>(d) * H + h) * W + w;
              dot = dot_row<VT, NV>(reinterpret_cast<const u32x4*>(vhead + grow * row_stride), go);
            }
            const A wd = dd ? ld[p] : A(1) - ld[p], wh = dh ? lh[p] : A(1) - lh[p], ww = dw ? lw[p] : A(1) - lw[p];
            pa += (wd * wh * ww) * dot;
            px += (dw ? dot : -dot) * (wd * wh);
            py += (dh ? dot : -dot) * (wd * ww);
            pz += (dd ? dot : -dot) * (wh * ww);
          }
        }
      }
      // the two channel halves of the query sit in neighbouring lanes
      pa += __shfl_xor(pa, 1, 64); px += __shfl_xor(px, 1, 64);
      py += __shfl_xor(py, 1, 64); pz += __shfl_xor(pz, 1, 64);
      res_attn[p] = pa;
      res_loc[3 * p] = px * aw[p] * static_cast<A>(W);
      res_loc[3 * p + 1] = py * aw[p] * static_cast<A>(H);
      res_loc[3 * p + 2] = pz * aw[p] * static_cast<A>(D);
    }
    if (live && half == 0) {
      const long j0 = item * LP + l * P;
      if (bin_count != nullptr) {
#pragma unroll
        for (int p = 0; p < P; ++p) bin_rank[j0 + p] = rank[p];
      }
#pragma unroll
      for (int p = 0; p < P; ++p) Elem<LT>::st(grad_attn + j0 + p, res_attn[p]);
#pragma unroll
      for (int i = 0; i < 3 * P; ++i) Elem<LT>::st(grad_loc + 3 * j0 + i, res_loc[i]);
    }
    __syncthreads();
  }
}

}  // namespace transoar
