// grad_value of the coarse levels on the matrix cores.
//
// Same unit of work as msda3d_bwd_value_cells (msda3d_tile.hpp): a wave owns kCmCellChunk consecutive SORTED
// points of the coarse levels of one (batch, head) slab and visits each exactly once.  The points of 7
// consecutive cells along w (one (cd, ch) cell row) touch 2 x 2 x 8 = 32 voxels; their contribution
//     dV[voxel, c] += sum_p  weight[voxel, p] * grad_out[item(p), c]
// is the product  A[32 voxels x K points] . B[K points x 64 channels]  of a weight matrix with two non-zeros
// per row and point (the point's dw = 0 / 1 corners of that voxel row's (dd, dh)) and the gathered grad_out
// rows: 16 points per MFMA K-step instead of 16 x (1 load + 8 FMA + scalar bookkeeping) in the lane = channel
// walk.  The weights (fp32, rebuilt from the 16-byte PointR16 record when it is staged) go to the matrix cores as
// hi + lo 16-bit halves (two MFMAs, 2^-16 relative), the grad_out rows are 16-bit already: fp32 accumulation of
// exact products as before.
//
// A-operand row i = wv + 8 * (2 dd + dh): voxel (cd - 1 + dd, ch - 1 + dh, 7 kw - 1 + wv) of window kw (cells
// 7 kw .. 7 kw + 6 of the row, cell = floor + 1 per axis).  Row wv receives the dw = 0 weight of the points of
// cell wv and the dw = 1 weight of the points of cell wv - 1: because the points are sorted by cell, both are
// index ranges [o1, o2) and [o0, o1) of the sorted list -- three offsets per lane and window, two unsigned
// compares per point.  In the D layout a register holds one (dd, dh) and two voxels (one per half-wave) x 32
// channels: a window is flushed with row-wide atomics into the fp32 scratch rows, like the per-cell flush of
// the scalar walk (16 lane-atomics per point at 30 points per cell, the same as there).
#pragma once
#include "msda3d_common.hpp"
#include "msda3d_mma.hpp"
#include "msda3d_tile.hpp"

namespace transoar {

constexpr int kCellsWindow = 7;        // cells per window: 8 voxels along w
constexpr int kCmCellChunk = 1024;     // sorted points per wave of the coarse walk (256: 2.41, 512: 2.35, 1024: 2.31, 2048: 2.44 ms per backward call)
constexpr int kCmRowPitch = 144;       // bytes per staged grad_out row (128 + 16: spreads the banks, as kMmaVP)

// Records from per-point ranks: the fill pass of the problems whose grad_loc / grad_attn kernel is not
// msda3d_bwd_query_mma (queries that are not the pyramid's voxels), which writes the records itself.
template <typename LT>
__global__ __launch_bounds__(256) void msda3d_cell_fill_r16(
    const LT* __restrict__ loc, const LT* __restrict__ attn, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const int* __restrict__ offset, const int* __restrict__ rank,
    PointR16* __restrict__ recs, int M, int L, int Lq, int P, long n_points) {
  const long j = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (j >= n_points) return;
  const int rk = rank[j];
  if (rk < 0) return;
  PointRec<float> rec;
  int item;
  const int bin = point_bin<LT, float>(loc, attn, shapes, lsi, j, M, L, Lq, P, &rec, &item);
  recs[offset[bin] + rk] = make_point_r16(rec.a, item, rec.ld, rec.lh, rec.lw);
}

// The K loop shared by the two grad_value kernels: points [t, t_end) of the sorted list, 16 per step, against
// the lane's voxel row (dw = 1 weight for points in [o1 - n1, o1), dw = 0 weight for [o1, o1 + n0)).  vrow / wrec:
// this wave's private staging areas (16 grad_out rows, 16 records).  Indices are clamped to `last`.
template <typename VT>
__device__ __forceinline__ void cell_run_mma(
    const VT* __restrict__ grad_out, const PointR16* __restrict__ recs,
    int t, int t_end, int last, int o0, int o1, unsigned n0, unsigned n1, unsigned char* vrow, float* wrec, int lane,
    f32x16& acc0, f32x16& acc1) {
  constexpr int C = kTileC;
  const int kg = lane >> 5, g4 = (lane & 31) >> 3;
  const int st_row = lane >> 2, st_q = lane & 3;        // staging: 4 lanes per grad_out row, 2 x 16 bytes each
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  // one K-step = 16 points: their records (256 contiguous bytes; the four lanes that fetch a point's grad_out row
  // all read its whole 16-byte record -- one request per quad) and their grad_out rows.  The row address depends on
  // the record, i.e. two memory latencies in series per step: records are requested THREE steps ahead and rows TWO,
  // so that neither wait is exposed once a run is under way (one step ahead for both, a step paid the record's
  // latency in full: the walks were 65 % wait cycles).
  auto load_rec = [&](int s) -> u32x4 {
    return *reinterpret_cast<const u32x4*>(recs + min(s + (lane >> 2), last));
  };
  auto load_rows = [&](const u32x4& rec, u32x4 (&rows)[2]) {
    const long item = static_cast<int>(rec.y);
    const u32x4* src = reinterpret_cast<const u32x4*>(grad_out + item * C);
    rows[0] = src[st_q];
    rows[1] = src[4 + st_q];
  };
  u32x4 rec0 = load_rec(t), rec1 = rec0, rec2 = rec0;       // records of steps s, s + 16, s + 32
  if (t + 16 < t_end) rec1 = load_rec(t + 16);
  if (t + 32 < t_end) rec2 = load_rec(t + 32);
  u32x4 row0[2], row1[2];                                    // grad_out rows of steps s, s + 16
  load_rows(rec0, row0);
  row1[0] = row0[0]; row1[1] = row0[1];
  if (t + 16 < t_end) load_rows(rec1, row1);
  for (int s = t; s < t_end; s += 16) {
    if ((lane & 2) == 0) {
      // lanes 0 / 1 of a quad expand the record into the weights of dd = 0 / 1: [t (1 - lw), t lw] per dh
      float ld, lh, lw;
      point_r16_fracs(rec0.z, rec0.w, ld, lh, lw);
      const float td = __uint_as_float(rec0.x) * ((lane & 1) ? ld : 1.f - ld);
      const float tb = td * lh, ta = td - tb;
      const float ya = ta * lw, yb = tb * lw;
      *reinterpret_cast<float4*>(wrec + (lane >> 2) * 8 + (lane & 1) * 4) = float4{ta - ya, ya, tb - yb, yb};
    }
    *reinterpret_cast<u32x4*>(vrow + st_row * kCmRowPitch + st_q * 16) = row0[0];
    *reinterpret_cast<u32x4*>(vrow + st_row * kCmRowPitch + 64 + st_q * 16) = row0[1];
    // rotate the pipeline: rows of step s + 32 (their records arrived a step ago), records of step s + 48
    rec0 = rec1; rec1 = rec2;
    row0[0] = row1[0]; row0[1] = row1[1];
    if (s + 32 < t_end) load_rows(rec1, row1);
    if (s + 48 < t_end) rec2 = load_rec(s + 48);
    // ---- A: this lane's voxel row x points s + 8 kg .. + 7
    float wa[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float2 pair = *reinterpret_cast<const float2*>(wrec + (8 * kg + e) * 8 + 2 * g4);
      const unsigned p = static_cast<unsigned>(s + 8 * kg + e);
      wa[e] = (p - static_cast<unsigned>(o1)) < n0 ? pair.x : ((p - static_cast<unsigned>(o0)) < n1 ? pair.y : 0.f);
    }
    s16x8 ahi, alo;
    Mma<VT>::split(wa, ahi, alo);
    // ---- B = grad_out rows, points along K: transposing reads (the layout of the forward's A = V^T)
    const unsigned char* bbase = vrow + (8 * kg + ((lane & 15) >> 2)) * kCmRowPitch + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 b00 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(bbase));
    const s16x4 b01 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(bbase + 4 * kCmRowPitch));
    const s16x4 b10 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(bbase + 64));
    const s16x4 b11 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(bbase + 4 * kCmRowPitch + 64));
    const s16x8 b0 = __builtin_shufflevector(b00, b01, 0, 1, 2, 3, 4, 5, 6, 7);
    const s16x8 b1 = __builtin_shufflevector(b10, b11, 0, 1, 2, 3, 4, 5, 6, 7);
    acc0 = Mma<VT>::mfma(ahi, b0, acc0);
    acc1 = Mma<VT>::mfma(ahi, b1, acc1);
    acc0 = Mma<VT>::mfma(alo, b0, acc0);
    acc1 = Mma<VT>::mfma(alo, b1, acc1);
  }
}

template <typename VT>
__global__ __launch_bounds__(256, 4) void msda3d_bwd_value_cells_mma(
    const VT* __restrict__ grad_out, const int* __restrict__ offset,
    const PointR16* __restrict__ recs,
    float* __restrict__ scratch, int cells_per_slab, int n_slabs, int M, const CoarseLevels* __restrict__ cl_p,
    const BrickOrder* __restrict__ order_p) {
  const CoarseLevels& cl = *cl_p;
  const BrickOrder& order = *order_p;
  constexpr int C = kTileC;
  __shared__ __attribute__((aligned(16))) unsigned char lds_rows[4][16 * kCmRowPitch];
  __shared__ __attribute__((aligned(16))) float lds_recs[4][16 * 8];

  const int lane = threadIdx.x & 63, wave_in_wg = uniform(threadIdx.x >> 6);
  // (an XCD-contiguous chunk order -- neighbouring cells, whose points share grad_out rows, on ONE L2 -- measured
  // SLOWER: 0.39 -> 0.43 ms; neighbouring chunks also flush into the same scratch rows, and their atomics then collide)
  const int wid = uniform(static_cast<int>(blockIdx.x) * 4 + wave_in_wg);
  const int slab = wid / cl.chunks_per_slab, chunk = wid - slab * cl.chunks_per_slab;
  if (slab >= n_slabs) return;
  const int* off = offset + static_cast<long>(slab) * cells_per_slab;
  const int t0 = off[cl.cell_start] + chunk * kCmCellChunk;
  const int end = min(t0 + kCmCellChunk, off[cells_per_slab]);
  if (t0 >= end) return;
  const int b = slab / M, m = slab - b * M;
  unsigned char* vrow = lds_rows[wave_in_wg];
  float* wrec = lds_recs[wave_in_wg];

  // cell of the first point: last cell whose list starts at or before t0
  int lo = cl.cell_start, hi = cells_per_slab;          // off[lo] <= t0 < off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= t0) lo = mid; else hi = mid;
  }
  int cell = lo;

  const int i = lane & 31, kg = lane >> 5;
  const int wv = i & 7;                                 // A row i = wv + 8 * (2 dd + dh)

  int t = t0;
  while (t < end) {
    while (t >= off[cell + 1]) ++cell;                  // skip empty cells
    // ---- window of the cell: level, (cd, ch, cw) = floor + 1 per axis
    int l = cl.first, base = cl.cell_start;
    for (int q = cl.first; q < order.L - 1; ++q) {
      const int n = (order.D[q] + 1) * (order.H[q] + 1) * (order.W[q] + 1);
      if (cell >= base + n && l == q) { base += n; l = q + 1; }
    }
    const int D = order.D[l], H = order.H[l], W = order.W[l];
    const int local = cell - base;
    const int cd = local / ((H + 1) * (W + 1));
    const int rr = local - cd * (H + 1) * (W + 1);
    const int ch = rr / (W + 1), cw = rr - ch * (W + 1);
    const int kw = cw / kCellsWindow;
    const int cell0 = cell - (cw - kw * kCellsWindow);
    const int ncell = min(kCellsWindow, W + 1 - kw * kCellsWindow);
    const int tile_end = min(end, off[cell0 + ncell]);
    // the lane's two source ranges of the sorted list, cut to this chunk
    const int o0 = min(max(off[cell0 + min(max(wv - 1, 0), ncell)], t), tile_end);
    const int o1 = min(max(off[cell0 + min(wv, ncell)], t), tile_end);
    const int o2 = min(max(off[cell0 + min(wv + 1, ncell)], t), tile_end);
    const unsigned n1 = static_cast<unsigned>(o1 - o0), n0 = static_cast<unsigned>(o2 - o1);

    f32x16 acc0, acc1;
    cell_run_mma<VT>(grad_out, recs, t, tile_end, end - 1, o0, o1, n0, n1, vrow, wrec, lane, acc0, acc1);

    // ---- flush: register r = (dd, dh) r >> 2, voxel wv = (r & 3) + 4 * (lane >> 5); lanes = 32 channels
    const int c_lo = max(cw - kw * kCellsWindow, 0);                        // first cell of the window seen by this chunk
    int c_hi = c_lo;                                                        // last one: the cell of point tile_end - 1
    while (c_hi + 1 < ncell && off[cell0 + c_hi + 1] < tile_end) ++c_hi;
    const long row0 = static_cast<long>(b) * cl.rows + (order.start[l] - cl.row_start);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = r >> 3, dh = (r >> 2) & 1;
      const int vd = cd - 1 + dd, vh = ch - 1 + dh;
      if (vd < 0 || vd >= D || vh < 0 || vh >= H) continue;               // wave-uniform
      const int wvr = (r & 3) + 4 * kg;
      const int vw = kw * kCellsWindow - 1 + wvr;
      if (wvr >= c_lo && wvr <= c_hi + 1 && vw >= 0 && vw < W) {
        float* dst = scratch + ((row0 + (vd * H + vh) * W + vw) * M + m) * C + i;
        atomic_accum(dst, acc0[r]);
        atomic_accum(dst + 32, acc1[r]);
      }
    }
    t = tile_end;
    cell = cell0 + ncell - 1;                 // the skip loop moves on from the window's last cell
    if (t < end) ++cell;
  }
}

// ---------------------------------------------------------------------------
// Fine levels: the brick-owner schedule of msda3d_bwd_value_tile (one workgroup per 4x4x8 brick, fp32 tile in
// LDS, 25 (d, h) cell rows in four parity classes) with the walk of a cell row on the matrix cores: the 9 cells
// of a row touch the brick's 8 voxels along w, row wv takes the dw = 1 weights of cell wv and the dw = 0 weights
// of cell wv + 1, and the 2 x 2 x 8 result is added to the tile once per row.
// ---------------------------------------------------------------------------
template <typename VT>
__global__ __launch_bounds__(kBrickThreads) void msda3d_bwd_value_tile_mma(
    const VT* __restrict__ grad_out, const int* __restrict__ offset,
    const PointR16* __restrict__ recs,
    VT* __restrict__ grad_value, int cells_per_slab, int S, int M, int fine_bricks, long n_wg, const BrickOrder* __restrict__ order_p) {
  const BrickOrder& order = *order_p;
  constexpr int C = kTileC;
  constexpr int NW = kBrickThreads / 64;
  __shared__ float tile[kBrickSlots * C];
  __shared__ int row_off[kCellRows][kCellsPerRow + 3];
  __shared__ __attribute__((aligned(16))) unsigned char lds_rows[NW][16 * kCmRowPitch];
  __shared__ __attribute__((aligned(16))) float lds_recs[NW][16 * 8];

  // (XCD-contiguous brick order measured 3 % slower: 0.571 -> 0.588 ms)
  const long wg = blockIdx.x;
  if (wg >= n_wg) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int m = static_cast<int>(wg % M);
  const long t1 = wg / M;
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

  const int i = lane & 31, kg = lane >> 5, wv = i & 7;
  unsigned char* vrow = lds_rows[wave];
  float* wrec = lds_recs[wave];
  for (int phase = 0; phase < 4; ++phase) {
    if (phase) __syncthreads();
    const int pd = phase >> 1, ph = phase & 1;
    const int n_h = (kBrickH + 2 - ph) / 2;                       // rows of this parity along h
    const int n_rows = ((kBrickD + 2 - pd) / 2) * n_h;
    for (int ri = wave; ri < n_rows; ri += NW) {
      const int rd = 2 * (ri / n_h) + pd, rh = 2 * (ri % n_h) + ph;
      const int row = rd * (kBrickH + 1) + rh;
      const bool d_ok[2] = {rd >= 1, rd <= kBrickD - 1 && d0 + rd < D};
      const bool h_ok[2] = {rh >= 1, rh <= kBrickH - 1 && h0 + rh < H};
      const int beg = uniform(row_off[row][0]), end = uniform(row_off[row][kCellsPerRow]);
      if (beg >= end) continue;
      const int o0 = row_off[row][wv], o1 = row_off[row][wv + 1], o2 = row_off[row][wv + 2];
      f32x16 acc0, acc1;
      cell_run_mma<VT>(grad_out, recs, beg, end, end - 1, o0, o1, static_cast<unsigned>(o2 - o1),
                       static_cast<unsigned>(o1 - o0), vrow, wrec, lane, acc0, acc1);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dd = r >> 3, dh = (r >> 2) & 1;
        if (!d_ok[dd] || !h_ok[dh]) continue;                               // wave-uniform
        const int wvr = (r & 3) + 4 * kg;
        if (w0 + wvr < W) {
          float* dst = tile + (((rd - 1 + dd) * kBrickH + (rh - 1 + dh)) * kBrickW + wvr) * C + i;
          dst[0] += acc0[r];
          dst[32] += acc1[r];
        }
      }
    }
  }
  __syncthreads();

  // tile -> grad_value rows (every voxel of the brick that exists in the level)
  for (int slot = wave; slot < kBrickSlots; slot += NW) {
    const int d = d0 + (slot >> 5), h = h0 + ((slot >> 3) & 3), w = w0 + (slot & 7);
    if (d >= D || h >= H || w >= W) continue;
    const long row = (static_cast<long>(b) * S + order.start[l] + (d * H + h) * W + w) * M + m;
    Elem<VT>::st(grad_value + row * C + lane, tile[slot * C + lane]);
  }
}

}  // namespace transoar
