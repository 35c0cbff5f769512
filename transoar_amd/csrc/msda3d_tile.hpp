// grad_value, brick-owner schedule: one workgroup owns the 4x4x8 voxels of a
// brick (for one batch element and head), keeps their 64-channel rows as an
// fp32 tile in LDS, and walks ONCE over the sorted sampling points of every
// cell that can touch the brick (a 5x5x9 cell box).  A point adds to the (up
// to 8) corner voxels that lie inside the brick; the other corners belong to
// the neighbouring bricks, whose workgroups visit the same point.
//
// Against the voxel-stationary pull (msda3d_scatter.hpp), which visits every
// point from each of its 8 corner voxels: a point and its grad_out row are
// fetched 225/128 = 1.76 times instead of 8, nothing is reduced across lanes,
// and the grad_value rows are written once, without atomics in global memory.
//
// lane = channel.  Everything about a point is wave-uniform (scalar loads,
// scalar control flow); the only vector work per point is one 2/4-byte load of
// the grad_out row and 8 fused multiply-adds into the 8 corner accumulators,
// which are flushed to the tile (ds_add_f32) when the walk leaves the cell.
// Points use the PointW8 record (the 8 corner weights, attention included)
// written by msda3d_cell_fill_w8.
#pragma once
#include "msda3d_common.hpp"
#include "msda3d_scatter.hpp"

namespace transoar {

template <typename A> struct alignas(16) PointW8 { A w[8]; };   // corner k = dd*4 + dh*2 + dw

constexpr int kTileC = 64;                       // channels per head this kernel is built for
constexpr int kCellRows = (kBrickD + 1) * (kBrickH + 1);   // (d,h) cell rows of the box
constexpr int kCellsPerRow = kBrickW + 1;

template <typename LT, typename A>
__global__ __launch_bounds__(256) void msda3d_cell_fill_w8(
    const LT* __restrict__ loc, const LT* __restrict__ attn, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const int* __restrict__ offset, const int* __restrict__ rank,
    PointW8<A>* __restrict__ recs, int* __restrict__ rec_item, int M, int L, int Lq, int P,
    long n_points) {
  const long j = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (j >= n_points) return;
  const int rk = rank[j];
  if (rk < 0) return;
  PointRec<A> rec;
  int item;
  const int bin = point_bin<LT, A>(loc, attn, shapes, lsi, j, M, L, Lq, P, &rec, &item);
  const int pos = offset[bin] + rk;
  PointW8<A> out;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    out.w[k] = ((k & 4) ? rec.ld : A(1) - rec.ld) * ((k & 2) ? rec.lh : A(1) - rec.lh) *
               ((k & 1) ? rec.lw : A(1) - rec.lw) * rec.a;
  recs[pos] = out;
  rec_item[pos] = item;
}

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <typename VT>
__global__ __launch_bounds__(kBrickThreads) void msda3d_bwd_value_tile(
    const VT* __restrict__ grad_out, const int* __restrict__ offset,
    const PointW8<float>* __restrict__ recs, const int* __restrict__ rec_item,
    VT* __restrict__ grad_value, int cells_per_slab, int S, int M, int fine_bricks, long n_wg, const BrickOrder* __restrict__ order_p) {
  const BrickOrder& order = *order_p;      // device-resident launch constants (msda3d.hip: device_const)
  constexpr int C = kTileC;
  constexpr int U = 4;                            // weight records per scalar-load round
  constexpr int G = 16;                           // grad_out rows in flight per wave
  __shared__ float tile[kBrickSlots * C];
  __shared__ int row_off[kCellRows][kCellsPerRow + 3];

  // plain round-robin over the XCDs: the bricks are ordered by level and the levels differ 500x
  // in points per brick, so a contiguous range per XCD would leave the heavy ones on one XCD
  const long wg = blockIdx.x;
  if (wg >= n_wg) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int m = static_cast<int>(wg % M);
  const long t1 = wg / M;
  // coarser levels (fewer bricks, longer point lists) first: no tail of heavy workgroups
  const int brick = fine_bricks - 1 - static_cast<int>(t1 % fine_bricks);
  const int b = static_cast<int>(t1 / fine_bricks);

  int l = 0;
  for (int t = 1; t < order.L; ++t) l += (brick * kBrickSlots >= order.pad_start[t]) ? 1 : 0;
  const int bk = (brick * kBrickSlots - order.pad_start[l]) >> 7;
  const int D = order.D[l], H = order.H[l], W = order.W[l];
  const int bw = bk % order.nbw[l];
  const int r2 = bk / order.nbw[l];
  const int d0 = (r2 / order.nbh[l]) * kBrickD, h0 = (r2 % order.nbh[l]) * kBrickH, w0 = bw * kBrickW;
  int cell_start = 0;
  for (int t = 0; t < l; ++t) cell_start += (order.D[t] + 1) * (order.H[t] + 1) * (order.W[t] + 1);
  const int bin0 = (b * M + m) * cells_per_slab + cell_start;
  const int cw_last = min(w0 + kBrickW, W);       // last cell (index along w) that touches the brick

  for (int i = tid; i < kBrickSlots * C / 4; i += kBrickThreads)
    reinterpret_cast<float4*>(tile)[i] = float4{0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < kCellRows * (kCellsPerRow + 1); i += kBrickThreads) {
    const int row = i / (kCellsPerRow + 1), j = i - row * (kCellsPerRow + 1);
    const int cd = d0 + row / (kBrickH + 1), ch = h0 + row % (kBrickH + 1);   // cell index = floor + 1
    int v = 0;
    if (cd <= D && ch <= H) v = offset[bin0 + (cd * (H + 1) + ch) * (W + 1) + min(w0 + j, cw_last + 1)];
    row_off[row][j] = v;
  }
  __syncthreads();

  // Cell rows whose (rd, rh) differ by 2 or more on an axis touch disjoint voxels, so the 25
  // rows are walked in four parity classes with a barrier in between; inside a class each row
  // belongs to one wave and the tile update is a plain read-add-write (LDS float atomics cost
  // 3.5x the whole walk when measured).
  for (int phase = 0; phase < 4; ++phase) {
  if (phase) __syncthreads();
  const int pd = phase >> 1, ph = phase & 1;
  const int n_h = (kBrickH + 2 - ph) / 2;                       // rows of this parity along h
  const int n_rows = ((kBrickD + 2 - pd) / 2) * n_h;
  for (int ri = wave; ri < n_rows; ri += kBrickThreads / 64) {
    const int rd = 2 * (ri / n_h) + pd, rh = 2 * (ri % n_h) + ph;
    const int row = rd * (kBrickH + 1) + rh;
    const bool d_ok[2] = {rd >= 1, rd <= kBrickD - 1 && d0 + rd < D};
    const bool h_ok[2] = {rh >= 1, rh <= kBrickH - 1 && h0 + rh < H};
    const int beg = uniform(row_off[row][0]), end = uniform(row_off[row][kCellsPerRow]);
    if (beg >= end) continue;

    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    int cwl = 0, boundary = uniform(row_off[row][1]);
    bool dirty = false;
    auto flush = [&]() {
      if (dirty) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int dd = k >> 2, dh = (k >> 1) & 1, dw = k & 1;
          const bool w_ok = dw ? (cwl <= kBrickW - 1 && w0 + cwl < W) : (cwl >= 1);
          if (d_ok[dd] && h_ok[dh] && w_ok) {
            const int slot = ((rd - 1 + dd) * kBrickH + (rh - 1 + dh)) * kBrickW + (cwl - 1 + dw);
            tile[slot * C + lane] += acc[k];
          }
          acc[k] = 0.f;
        }
      }
      dirty = false;
    };

    // 64 points per pass: their grad_out row indices arrive with one vector load (lane j ->
    // point t+j), after which the rows of G points are requested back to back -- the only
    // dependent step of the walk.  The weights come through scalar loads, whose addresses
    // depend on t alone.
    for (int t = beg; t < end; t += 64) {
      const int n = min(64, end - t);
      const int item_v = rec_item[min(t + lane, end - 1)];
      for (int j0 = 0; j0 < n; j0 += G) {
        float g[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
          const long item = __builtin_amdgcn_readlane(item_v, min(j0 + u, n - 1));
          g[u] = Elem<VT>::ld(grad_out + item * C + lane);
        }
#pragma unroll
        for (int u0 = 0; u0 < G; u0 += U) {
          PointW8<float> rc[U];
#pragma unroll
          for (int u = 0; u < U; ++u) rc[u] = recs[min(t + j0 + u0 + u, end - 1)];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int tt = t + j0 + u0 + u;
            if (tt < end) {
              while (tt >= boundary) {
                flush();
                ++cwl;
                boundary = uniform(row_off[row][cwl + 1]);
              }
#pragma unroll
              for (int k = 0; k < 8; ++k) acc[k] += rc[u].w[k] * g[u0 + u];
              dirty = true;
            }
          }
        }
      }
    }
    flush();
  }
  }
  __syncthreads();

  // tile -> grad_value rows (every voxel of the brick that exists in the level)
  for (int slot = wave; slot < kBrickSlots; slot += kBrickThreads / 64) {
    const int d = d0 + (slot >> 5), h = h0 + ((slot >> 3) & 3), w = w0 + (slot & 7);
    if (d >= D || h >= H || w >= W) continue;
    const long row = (static_cast<long>(b) * S + order.start[l] + (d * H + h) * W + w) * M + m;
    Elem<VT>::st(grad_value + row * C + lane, tile[slot * C + lane]);
  }
}

// ---------------------------------------------------------------------------
// Coarse levels.  A level with few voxels receives as many points as a fine one
// (every query samples every level), hundreds to thousands per cell, so there
// are too few bricks to fill the chip.  Here the unit of work is a fixed chunk
// of kCellChunk consecutive SORTED points of the coarse levels of one
// (batch, head) slab: a wave walks its chunk, keeps the 8 corner sums of the
// current cell in registers and, when the cell changes, adds them to an fp32
// scratch copy of the coarse rows with 8 row-wide atomics.  A point is visited
// exactly once.  msda3d_coarse_rows_store then writes the rows in the storage type.
// ---------------------------------------------------------------------------
constexpr int kCellChunk = 256;
constexpr long kCoarsePointsPerVoxel = 32;   // measured: 128 -> 32 moves level 1 of the flagship pyramid here (2.00 -> 1.90 ms serial)

struct CoarseLevels {
  int first;             // first coarse level (levels first..L-1)
  int cell_start;        // cell index inside a slab where level `first` begins
  int row_start;         // pyramid row where level `first` begins
  int rows;              // voxels of the coarse levels
  int chunks_per_slab;   // upper bound: every point falls into one level
};

template <typename VT>
__global__ __launch_bounds__(256) void msda3d_bwd_value_cells(
    const VT* __restrict__ grad_out, const int* __restrict__ offset,
    const PointW8<float>* __restrict__ recs, const int* __restrict__ rec_item,
    float* __restrict__ scratch, int cells_per_slab, int n_slabs, int M, const CoarseLevels* __restrict__ cl_p,
    const BrickOrder* __restrict__ order_p) {
  const CoarseLevels& cl = *cl_p;
  const BrickOrder& order = *order_p;
  constexpr int C = kTileC;
  constexpr int U = 4, G = 16;
  const int lane = threadIdx.x & 63;
  const int wid = uniform(static_cast<int>(blockIdx.x) * 4 + (threadIdx.x >> 6));
  const int slab = wid / cl.chunks_per_slab, chunk = wid - slab * cl.chunks_per_slab;
  if (slab >= n_slabs) return;
  const int* off = offset + static_cast<long>(slab) * cells_per_slab;
  const int t0 = off[cl.cell_start] + chunk * kCellChunk;
  const int end = min(t0 + kCellChunk, off[cells_per_slab]);
  if (t0 >= end) return;
  const int b = slab / M, m = slab - b * M;

  // cell of the first point: last cell whose list starts at or before t0
  int lo = cl.cell_start, hi = cells_per_slab;          // off[lo] <= t0 < off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= t0) lo = mid; else hi = mid;
  }
  int cell = lo, boundary = off[cell + 1];

  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  auto flush = [&]() {
    // decode the cell: level, then (cd, ch, cw) = floor + 1 per axis
    int l = cl.first, base = cl.cell_start;
    for (int t = cl.first; t < order.L - 1; ++t) {
      const int n = (order.D[t] + 1) * (order.H[t] + 1) * (order.W[t] + 1);
      if (cell >= base + n && l == t) { base += n; l = t + 1; }
    }
    const int D = order.D[l], H = order.H[l], W = order.W[l];
    const int local = cell - base;
    const int cd = local / ((H + 1) * (W + 1));
    const int r = local - cd * (H + 1) * (W + 1);
    const int ch = r / (W + 1), cw = r - ch * (W + 1);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int vd = cd - 1 + (k >> 2), vh = ch - 1 + ((k >> 1) & 1), vw = cw - 1 + (k & 1);
      if (vd >= 0 && vd < D && vh >= 0 && vh < H && vw >= 0 && vw < W) {
        const long row = (static_cast<long>(b) * cl.rows + (order.start[l] - cl.row_start) + (vd * H + vh) * W + vw) * M + m;
        atomic_accum(scratch + row * C + lane, acc[k]);
      }
      acc[k] = 0.f;
    }
  };

  for (int t = t0; t < end; t += 64) {
    const int n = min(64, end - t);
    const int item_v = rec_item[min(t + lane, end - 1)];
    for (int j0 = 0; j0 < n; j0 += G) {
      float g[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const long item = __builtin_amdgcn_readlane(item_v, min(j0 + u, n - 1));
        g[u] = Elem<VT>::ld(grad_out + item * C + lane);
      }
#pragma unroll
      for (int u0 = 0; u0 < G; u0 += U) {
        PointW8<float> rc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) rc[u] = recs[min(t + j0 + u0 + u, end - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int tt = t + j0 + u0 + u;
          if (tt < end) {
            if (tt >= boundary) {
              flush();
              do { ++cell; boundary = off[cell + 1]; } while (tt >= boundary);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += rc[u].w[k] * g[u0 + u];
          }
        }
      }
    }
  }
  flush();
}

// scratch (N, rows, M, C) fp32 -> the coarse rows of grad_value (N, S, M, C)
template <typename VT>
__global__ __launch_bounds__(256) void msda3d_coarse_rows_store(const float* __restrict__ scratch,
                                                                VT* __restrict__ grad_value, int S, int M,
                                                                int cl_rows, int cl_row_start, long n) {
  struct { int rows, row_start; } cl{cl_rows, cl_row_start};
  const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;    // one float4 each
  if (i >= n) return;
  constexpr int V4 = kTileC / 4;
  const long row = i / V4;                                             // (b*rows + r)*M + m
  const int c4 = static_cast<int>(i - row * V4);
  const int m = static_cast<int>(row % M);
  const long br = row / M;
  const int r = static_cast<int>(br % cl.rows);
  const long b = br / cl.rows;
  const float4 v = reinterpret_cast<const float4*>(scratch)[i];
  VT* dst = grad_value + ((b * S + cl.row_start + r) * M + m) * kTileC + c4 * 4;
  Elem<VT>::st(dst, v.x);
  Elem<VT>::st(dst + 1, v.y);
  Elem<VT>::st(dst + 2, v.z);
  Elem<VT>::st(dst + 3, v.w);
}

}  // namespace transoar
