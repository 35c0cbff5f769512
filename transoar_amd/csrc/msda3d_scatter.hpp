// grad_value of MSDeformAttn-3D without atomics (gfx950).
//
// The reference scatters w_k*a*g into grad_value with one atomicAdd per
// channel per corner (ms_deform_im2col_cuda.cuh:162-232): 11.5e9 float atomics
// at the flagship shape, ~120 ms on MI355X when done that way.  Here the
// scatter is turned into a gather over the transposed sampling graph:
//
//   1. cell_count   every valid sampling point is binned by the cell
//                   (d0,h0,w0) its 8 corners hang off; one int atomic per
//                   POINT (not per channel*corner) returns its rank in the bin
//   2. scan         exclusive prefix sum of the bin counts
//   3. cell_fill    each point drops a 16-byte record {ld,lh,lw,a} and its
//                   item index at offset[bin]+rank  -> points sorted by cell
//   4. pull         one wave per grad_value row (b, voxel, head): the voxel is
//                   corner delta of the 8 cells (voxel - delta); walk their
//                   record lists, gather the grad_out rows (16 B per lane,
//                   whole rows, like the forward gather) and accumulate in
//                   registers; one plain coalesced store per row.
//
// Bins use a padded grid (D+1)(H+1)(W+1) per level because the low corner of
// a point may be -1 on any axis.
#pragma once
#include "msda3d_common.hpp"

namespace transoar {

template <typename A> struct alignas(16) PointRec { A ld, lh, lw, a; };

constexpr int kPullUnroll = 4;
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;                       // per thread
constexpr int kScanTile = kScanThreads * kScanItems;

struct LevelGrid {
  int D, H, W, start, cell_start;
};

// geometry of level l and the total padded cell count S' (uniform loops, L<=8)
__device__ __forceinline__ LevelGrid level_grid(const int64_t* __restrict__ shapes,
                                                const int64_t* __restrict__ lsi, int l, int L,
                                                int* total_cells) {
  LevelGrid g{0, 0, 0, 0, 0};
  int acc = 0;
  for (int t = 0; t < L; ++t) {
    const int D = static_cast<int>(shapes[3 * t]), H = static_cast<int>(shapes[3 * t + 1]),
              W = static_cast<int>(shapes[3 * t + 2]);
    if (t == l) {
      g.D = D; g.H = H; g.W = W;
      g.start = static_cast<int>(lsi[t]);
      g.cell_start = acc;
    }
    acc += (D + 1) * (H + 1) * (W + 1);
  }
  *total_cells = acc;
  return g;
}

// bin (or -1) and fractions of point j (global point index over N*Lq*M*L*P)
template <typename LT, typename A>
__device__ __forceinline__ int point_bin(const LT* __restrict__ loc, const LT* __restrict__ attn,
                                         const int64_t* __restrict__ shapes,
                                         const int64_t* __restrict__ lsi, long j, int M, int L,
                                         int Lq, int P, PointRec<A>* rec, int* item_out) {
  const int LP = L * P;
  const int item = static_cast<int>(j / LP);
  const int lp = static_cast<int>(j - static_cast<long>(item) * LP);
  const int l = lp / P;
  const int m = item % M;
  const int b = (item / M) / Lq;
  int cells;
  const LevelGrid g = level_grid(shapes, lsi, l, L, &cells);
  const A w_im = pixel_coord(static_cast<A>(Elem<LT>::ld(loc + 3 * j)), g.W);
  const A h_im = pixel_coord(static_cast<A>(Elem<LT>::ld(loc + 3 * j + 1)), g.H);
  const A d_im = pixel_coord(static_cast<A>(Elem<LT>::ld(loc + 3 * j + 2)), g.D);
  *item_out = item;
  if (!(d_im > A(-1) && h_im > A(-1) && w_im > A(-1) && d_im < g.D && h_im < g.H && w_im < g.W))
    return -1;
  const A fd = floor(d_im), fh = floor(h_im), fw = floor(w_im);
  rec->ld = d_im - fd;
  rec->lh = h_im - fh;
  rec->lw = w_im - fw;
  rec->a = static_cast<A>(Elem<LT>::ld(attn + j));
  const int cd = static_cast<int>(fd) + 1, ch = static_cast<int>(fh) + 1, cw = static_cast<int>(fw) + 1;
  return (b * M + m) * cells + g.cell_start + (cd * (g.H + 1) + ch) * (g.W + 1) + cw;
}

template <typename LT, typename A>
__global__ __launch_bounds__(256) void msda3d_cell_count(
    const LT* __restrict__ loc, const LT* __restrict__ attn, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, int* __restrict__ count, int* __restrict__ rank, int M, int L,
    int Lq, int P, long n_points) {
  const long j = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (j >= n_points) return;
  PointRec<A> rec;
  int item;
  const int bin = point_bin<LT, A>(loc, attn, shapes, lsi, j, M, L, Lq, P, &rec, &item);
  rank[j] = bin < 0 ? -1 : atomicAdd(count + bin, 1);
}

template <typename LT, typename A>
__global__ __launch_bounds__(256) void msda3d_cell_fill(
    const LT* __restrict__ loc, const LT* __restrict__ attn, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const int* __restrict__ offset, const int* __restrict__ rank,
    PointRec<A>* __restrict__ recs, int* __restrict__ rec_item, int M, int L, int Lq, int P,
    long n_points) {
  const long j = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (j >= n_points) return;
  const int rk = rank[j];
  if (rk < 0) return;
  PointRec<A> rec;
  int item;
  const int bin = point_bin<LT, A>(loc, attn, shapes, lsi, j, M, L, Lq, P, &rec, &item);
  const int pos = offset[bin] + rk;
  recs[pos] = rec;
  rec_item[pos] = item;
}

// ---- exclusive scan over n ints, in place, three passes --------------------
__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
  __shared__ int wave_sums[kScanThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  int base = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; ++w) {
    const int s = wave_sums[w];
    if (w < wave) base += s;
    all += s;
  }
  __syncthreads();
  *total = all;
  return base + incl - v;
}

__global__ __launch_bounds__(kScanThreads) void msda3d_scan_tiles(int* __restrict__ data,
                                                                  int* __restrict__ tile_sums, int n) {
  const int tile0 = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  int v[kScanItems], sum = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    v[i] = tile0 + i < n ? data[tile0 + i] : 0;
    sum += v[i];
  }
  int total;
  int run = block_exclusive_scan(sum, &total);
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    if (tile0 + i < n) data[tile0 + i] = run;
    run += v[i];
  }
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// one block: exclusive scan of the tile sums (n_tiles may exceed the block)
__global__ __launch_bounds__(kScanThreads) void msda3d_scan_tile_sums(int* __restrict__ tile_sums,
                                                                      int n_tiles) {
  int carry = 0;
  for (int base = 0; base < n_tiles; base += kScanThreads) {
    const int i = base + threadIdx.x;
    const int v = i < n_tiles ? tile_sums[i] : 0;
    int total;
    const int ex = block_exclusive_scan(v, &total);
    if (i < n_tiles) tile_sums[i] = carry + ex;
    carry += total;
  }
}

__global__ __launch_bounds__(kScanThreads) void msda3d_scan_add(int* __restrict__ data,
                                                                const int* __restrict__ tile_sums,
                                                                int n) {
  const int add = tile_sums[blockIdx.x];
  const int tile0 = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i)
    if (tile0 + i < n) data[tile0 + i] += add;
}

// ---- pull ------------------------------------------------------------------
// offset[] has one extra entry past the last bin (the scan covers n_bins+1).
template <typename VT, typename A, int LOG2_LPV>
__global__ __launch_bounds__(64 * kWavesPerBlock) void msda3d_bwd_value_pull(
    const VT* __restrict__ grad_out, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const int* __restrict__ offset,
    const PointRec<A>* __restrict__ recs, const int* __restrict__ rec_item,
    VT* __restrict__ grad_value, int S, int M, int C, int L, long n_units, long n_blocks,
    BrickOrder order, unsigned rec_bytes) {
  constexpr int VEC = Elem<VT>::VEC;
  constexpr int LPV = 1 << LOG2_LPV;
  constexpr int CPI = 64 / LPV;   // records handled per load instruction

  const long blk = xcd_contiguous_block(blockIdx.x, n_blocks);
  if (blk < 0) return;
  const int lane = threadIdx.x & 63;
  const long unit = __builtin_amdgcn_readfirstlane(
      static_cast<int>(blk * kWavesPerBlock + (threadIdx.x >> 6)));
  if (unit >= n_units) return;
  const long row = ordered_unit(order, unit, S, M);
  if (row < 0) return;
  const int m = static_cast<int>(row % M);
  const int bs = static_cast<int>(row / M);
  const int b = bs / S, s = bs - b * S;
  const int cv = lane & (LPV - 1);
  const int cg = lane >> LOG2_LPV;

  int l = 0;
  for (int t = 1; t < L; ++t) l += (s >= static_cast<int>(lsi[t])) ? 1 : 0;
  int cells;
  const LevelGrid g = level_grid(shapes, lsi, l, L, &cells);
  const int local = s - g.start;
  const int d = local / (g.H * g.W);
  const int hw = local - d * g.H * g.W;
  const int h = hw / g.W, w = hw - h * g.W;
  const int bin0 = (b * M + m) * cells + g.cell_start;

  A acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = A(0);

  // This voxel is corner k = (dd,dh,dw) of every point binned in cell (voxel - k).  The 8 record
  // lists are walked as ONE flattened sequence i = 0..total, CPI lane groups taking every CPI-th
  // record, so that the record/item loads of step i+CPI are in flight while step i's grad_out row
  // is being fetched (the dependent chain per step is one memory latency, not two).
  int beg[8], pre[9];
  pre[0] = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int dd = k >> 2, dh = (k >> 1) & 1, dw = k & 1;
    const int bin = bin0 + ((d - dd + 1) * (g.H + 1) + (h - dh + 1)) * (g.W + 1) + (w - dw + 1);
    beg[k] = offset[bin];
    pre[k + 1] = pre[k] + (offset[bin + 1] - beg[k]);
  }
  const int total = pre[8];
  auto locate = [&](int i, int& k, int& t) {
    k = 0;
    t = i + beg[0];
#pragma unroll
    for (int j = 1; j < 8; ++j)
      if (i >= pre[j]) { k = j; t = i - pre[j] + beg[j]; }
  };

  // kPullUnroll records per lane group are in flight together (records, then their grad_out
  // rows): the kernel is bound by bytes in flight per wave (Little's law), not by bandwidth.
  constexpr int U = kPullUnroll;
  const unsigned c_bytes = static_cast<unsigned>(C) * sizeof(VT);
  const unsigned lane_off = static_cast<unsigned>(cv * VEC * sizeof(VT));
  const __amdgpu_buffer_rsrc_t gr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<VT*>(grad_out), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<PointRec<A>*>(recs), 0, static_cast<int>(rec_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int*>(rec_item), 0, static_cast<int>(rec_bytes / (sizeof(PointRec<A>) / 4)), 0x00020000);
  for (int i0 = cg; i0 < total; i0 += CPI * U) {
    // branch-free: a lane group that has run out of records reads out of range (zeros) and
    // contributes weight 0, so all 2*U record loads issue back to back
    PointRec<A> rc[U];
    int item[U], kk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * CPI;
      const bool live = i < total;
      int t;
      locate(live ? i : 0, kk[u], t);
      const unsigned tu = live ? static_cast<unsigned>(t) : 0x0fffffffu;
      if constexpr (sizeof(A) == 4) {
        const u32x4 raw_rec = __builtin_amdgcn_raw_buffer_load_b128(rr, tu * 16u, 0, 0);
        rc[u].ld = __uint_as_float(raw_rec[0]);
        rc[u].lh = __uint_as_float(raw_rec[1]);
        rc[u].lw = __uint_as_float(raw_rec[2]);
        rc[u].a = __uint_as_float(raw_rec[3]);
      } else {
        rc[u] = live ? recs[t] : PointRec<A>{A(0), A(0), A(0), A(0)};
      }
      const int it = __builtin_amdgcn_raw_buffer_load_b32(ir, tu * 4u, 0, 0);
      item[u] = live ? it : -1;
    }
    u32x4 raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // a finished group reads nothing: out-of-range offset -> zeros from the buffer unit
      const unsigned off = item[u] >= 0 ? __umul24(static_cast<unsigned>(item[u]), c_bytes) + lane_off : 0xfffffff0u;
      raw[u] = __builtin_amdgcn_raw_buffer_load_b128(gr, off, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      A go[VEC];
      Elem<VT>::unpack(raw[u], go);
      const int k = kk[u];
      const A wt = ((k & 4) ? rc[u].ld : A(1) - rc[u].ld) * ((k & 2) ? rc[u].lh : A(1) - rc[u].lh) *
                   ((k & 1) ? rc[u].lw : A(1) - rc[u].lw) * rc[u].a;
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += wt * go[e];
    }
  }

#pragma unroll
  for (int mask = LPV; mask < 64; mask <<= 1) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] += xor_lanes(acc[e], mask);
  }
  if (cg == 0)
    *reinterpret_cast<u32x4*>(grad_value + row * C + cv * VEC) = Elem<VT>::pack(acc);
}

// fp32 accumulator -> 16-bit storage (generic path only)
template <typename VT>
__global__ __launch_bounds__(256) void msda3d_cast_rows(const float* __restrict__ src,
                                                        VT* __restrict__ dst, long n) {
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<long>(gridDim.x) * 256)
    Elem<VT>::st(dst + i, src[i]);
}

}  // namespace transoar
