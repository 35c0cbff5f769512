// Weight gradient of a 3x3x3 / pad 1 convolution (stride 1 or 2) with 40..64 channels on at least one side over
// 10^6..10^7 voxels: the 24 -> 48 / stride 2 and 48 -> 48 layers of stage 1 (backbones/encoder_blocks.py:28-51).
//   part[chunk][tap][co][ci] = sum over the chunk's dy voxels j:  dy[j][co] * x[S j + tap - 1][ci]
// conv3d_wgrad_kernel (the general voxel-major GEMM above) gives every (tap, channel tile) its own workgroups: x is
// fetched 27 times through L2 (4.2 GB for 48 -> 48 @ 80x80x128: 0.65 - 0.76 ms where the MFMAs take 0.14 and HBM 0.1).
// Here a workgroup owns ONE filter plane kd and ALL channels, and walks W-rows of dy with h fastest, keeping the three
// x rows (kh) of its plane in an LDS ring -- x and dy are fetched 3 times (once per plane) instead of 27:
//   unit     64 consecutive dy voxels of a W-row (b, jd, jh, jw0..jw0+63)
//   GT       [2][co tile][64 voxels][64 B]              the dy tile, double-buffered
//   XT       [ring][ci tile][columns][64 B]             x rows ih = S jh + kh - 1, columns iw = S jw0 - 1 + c (66 for
//            S = 1; 129 for S = 2, stored de-interleaved by parity so that the 16 voxels of a K step are consecutive rows
//            for every kw); ring of 4 (S = 1: one new row per unit) or 6 (S = 2: two) -- the next unit's rows are
//            written while this unit's are read, one barrier per unit
//   8 waves  = (co tile, ci tile) pairs x a split of the unit's K axis: with 2 x 2 channel tiles every pair has two
//            waves of 32 voxels each, with 2 tiles four waves of 16.  A wave multiplies its voxels for all 9 taps
//            (kh, kw) of the plane -- A and B fragments by ds_read_b64_tr_b16 (64-byte pitch: 4 rows = all 64 banks), 9
//            accumulators of 32 x 32 fp32 per wave, summed over the K split once at the end.  Every wave runs the same code.
//   grid     chunks x 3 planes, plane fastest; a persistent workgroup keeps its plane and its accumulators across all of
//            its tasks = (h chunk, W segment, (b, jd) slice), the slice fastest.
// (A first version gave every (plane, co tile, ci tile) its own workgroup of 4 waves: 0.60 ms on both layers, 2.5 GB of
// fetches at 4.1 TB/s -- the tiles of one task did not stay close enough in time to share L2.)
// Output = conv3d_wgrad_kernel's partial maps: conv3d_wgrad_reduce_kernel finishes.  Included by conv_gemm.hip.
#pragma once

template <int S> struct RingGeom {
  static constexpr int kCols = 64 * S + 2;                       // x columns a unit touches (66 / 130, the last of S = 2 unused)
  static constexpr int kPlane = S == 1 ? 0 : 66;                 // S = 2: odd columns start at row kPlane of the staged x row
  static constexpr int kRows = S == 1 ? 68 : 132;                // staged rows (positions) per x row and ci tile
  static constexpr int kRing = S == 1 ? 4 : 6;
  static constexpr int kNew = S;                                 // new x rows per unit
  __device__ static constexpr int position(int c) { return S == 1 ? c : (c & 1) * kPlane + (c >> 1); }
  __device__ static constexpr int tap_position(int kw) { return S == 1 ? kw : (kw == 1 ? kPlane : kw >> 1); }     // of voxel 0
};

template <int S, int CIT>          // CIT: ci tiles of 32 channels (1 or 2)
__global__ __launch_bounds__(512, 2) void conv3d_wgrad_ring_kernel(
    const unsigned short* __restrict__ DY, const unsigned short* __restrict__ X, float* __restrict__ part, int N, int SD, int SH, int SW,
    int MD, int MH, int MW, int Cin, int Cout, int tiles_co, int tiles_ci, int h_chunks, int h_chunk) {
  typedef RingGeom<S> G;
  constexpr int kTileB = G::kRows * 64;                           // bytes of one staged x row of one ci tile
  constexpr int kRowB = 2 * kTileB;                               // ... of both ci tiles = one ring slot
  constexpr int kPieces = 4 * CIT;                                // 16-byte pieces per x voxel (staging slots; pieces >= Cin / 8 are idle)
  constexpr int kXItems = (G::kNew * G::kCols * kPieces + 511) / 512;      // pieces of the new rows per thread
  __shared__ __attribute__((aligned(16))) unsigned char GT[2][2 * 64 * 64];
  __shared__ __attribute__((aligned(16))) unsigned char XT[G::kRing * kRowB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kd = static_cast<int>(blockIdx.x % 3), chunk = static_cast<int>(blockIdx.x / 3), chunks = static_cast<int>(gridDim.x / 3);
  const int pairs = tiles_co * tiles_ci;                          // 2 or 4 (host)
  const int ksplit = 8 / pairs, nks = 4 / ksplit;                 // waves per pair; K steps of 16 voxels per wave and unit
  const int pair = wave / ksplit, kpart = wave - pair * ksplit;
  const int cot = pair / tiles_ci, cit = pair - cot * tiles_ci;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int i = tid; i < 2 * 2 * 64 * 64 / 4; i += 512) reinterpret_cast<unsigned*>(&GT[0][0])[i] = 0u;      // channels beyond Cout / Cin stay zero
  for (int i = tid; i < G::kRing * kRowB / 4; i += 512) reinterpret_cast<unsigned*>(XT)[i] = 0u;

  // per-thread staging items: the same for every unit
  const int co8 = Cout >> 3, ci8 = Cin >> 3;
  const int g_vox = tid / co8, g_piece = tid - g_vox * co8;
  const bool g_valid = g_vox < 64;
  const unsigned g_lds = static_cast<unsigned>((g_piece >> 2) * 4096 + g_vox * 64 + (g_piece & 3) * 16);
  const unsigned g_goff = static_cast<unsigned>(g_vox * Cout + g_piece * 8);
  unsigned x_lds[kXItems], x_goff[kXItems];
  int x_meta[kXItems];                                 // new-row index | column << 8, or -1
  const long x_bias = static_cast<long>(Cin);          // column -1 of the first row: makes every item offset non-negative
#pragma unroll
  for (int k = 0; k < kXItems; ++k) {
    const int i = tid + 512 * k;
    const int q = i / kPieces, piece = i - q * kPieces, nr = q / G::kCols, c = q - nr * G::kCols;
    const bool ok = nr < G::kNew && piece < ci8 && (S == 1 || c < 129);
    x_lds[k] = static_cast<unsigned>((piece >> 2) * kTileB + G::position(c) * 64 + (piece & 3) * 16);
    x_goff[k] = static_cast<unsigned>(x_bias + (static_cast<long>(nr) * SW + (c - 1)) * Cin + piece * 8);
    x_meta[k] = ok ? (nr | (c << 8)) : -1;
  }
  u32x4 graw, xraw[kXItems];
  // rows of unit jh that the unit before it did not bring: kh = 3 - S .. 2, i.e. ih = S jh + 2 - S + nr
  auto fetch = [&](int b, int jd, int id, int jw0, int jh) {
    graw = u32x4{0u, 0u, 0u, 0u};
    if (g_valid) graw = *reinterpret_cast<const u32x4*>(DY + (((static_cast<long>(b) * MD + jd) * MH + jh) * MW + jw0) * Cout + g_goff);
    const int ih0 = S * jh + 2 - S;
    const unsigned short* xb = X + (((static_cast<long>(b) * SD + id) * SH + ih0) * SW + S * jw0) * Cin - x_bias;
#pragma unroll
    for (int k = 0; k < kXItems; ++k) {
      const int nr = x_meta[k] & 0xff, c = x_meta[k] >> 8;
      const bool ok = x_meta[k] >= 0 && ih0 + nr < SH && static_cast<unsigned>(S * jw0 + c - 1) < static_cast<unsigned>(SW);
      xraw[k] = u32x4{0u, 0u, 0u, 0u};
      if (ok) xraw[k] = *reinterpret_cast<const u32x4*>(xb + x_goff[k]);
    }
  };
  auto stage = [&](int jh) {
    if (g_valid) *reinterpret_cast<u32x4*>(&GT[jh & 1][g_lds]) = graw;
    const int ih0 = S * jh + 2 - S;
#pragma unroll
    for (int k = 0; k < kXItems; ++k)
      if (x_meta[k] >= 0) *reinterpret_cast<u32x4*>(XT + ((ih0 + (x_meta[k] & 0xff)) % G::kRing) * kRowB + x_lds[k]) = xraw[k];
  };
  // transposing fragment reads: lane supplies row (lane & 15) >> 2 (+ 8 kg) of a 4-row set and 4 channels
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int kg = lane >> 5;
  const int tr_off = (kpart * 16 * nks + 8 * kg + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  auto frag = [&](const unsigned char* p) -> s16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * 64));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  const int segs = MW / 64, nbd = N * MD;
  const long n_tasks = static_cast<long>(nbd) * segs * h_chunks;
  __syncthreads();
  for (long task = chunk; task < n_tasks; task += chunks) {
    const int bd = static_cast<int>(task % nbd), r_ = static_cast<int>(task / nbd);
    const int jw0 = (r_ % segs) * 64, hh = r_ / segs;
    const int b = bd / MD, jd = bd - b * MD;
    const int id = S * jd + kd - 1;
    if (static_cast<unsigned>(id) >= static_cast<unsigned>(SD)) continue;          // this plane lies in the padding (uniform)
    const int h_beg = hh * h_chunk, h_end = min(MH, h_beg + h_chunk);
    for (int jh = h_beg; jh < h_end; ++jh) {
      const int par = jh & 1;
      if (jh == h_beg) {
        // first unit of a task: its dy tile and all three x rows straight through (the barrier at the end of the previous
        // unit covers the buffers)
        if (g_valid) *reinterpret_cast<u32x4*>(&GT[par][g_lds]) =
            *reinterpret_cast<const u32x4*>(DY + (((static_cast<long>(b) * MD + jd) * MH + jh) * MW + jw0) * Cout + g_goff);
        for (int i = tid; i < 3 * G::kCols * kPieces; i += 512) {
          const int q = i / kPieces, piece = i - q * kPieces, kh = q / G::kCols, c = q - kh * G::kCols;
          if (piece >= ci8) continue;
          const int ih = S * jh + kh - 1, iw = S * jw0 + c - 1;
          if (S == 2 && c >= 129) continue;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (static_cast<unsigned>(ih) < static_cast<unsigned>(SH) && static_cast<unsigned>(iw) < static_cast<unsigned>(SW))
            v = *reinterpret_cast<const u32x4*>(X + (((static_cast<long>(b) * SD + id) * SH + ih) * SW + iw) * Cin + piece * 8);
          *reinterpret_cast<u32x4*>(XT + ((ih + G::kRing) % G::kRing) * kRowB + (piece >> 2) * kTileB + G::position(c) * 64 + (piece & 3) * 16) = v;
        }
        __syncthreads();
        if (jh + 1 < h_end) fetch(b, jd, id, jw0, jh + 1);
      }
      // unit jh + 1 (loaded one unit ago) -> the buffers nobody reads now; unit jh + 2 -> registers.  (Two register sets
      // and a distance of two units measured SLOWER: 0.60 / 0.69 ms against 0.48 / 0.51.)
      if (jh + 1 < h_end) stage(jh + 1);
      if (jh + 2 < h_end) fetch(b, jd, id, jw0, jh + 2);
      for (int ks = 0; ks < nks; ++ks) {
        const s16x8 a = frag(GT[par] + cot * 4096 + ks * 1024 + tr_off);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const unsigned char* xrow = XT + ((S * jh + kh - 1 + G::kRing) % G::kRing) * kRowB + cit * kTileB + ks * 1024 + tr_off;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = mfma(a, frag(xrow + G::tap_position(kw) * 64), acc[kh * 3 + kw]);
        }
      }
      __syncthreads();
    }
  }
  // ---- the K-split waves' accumulators -> one tile per (pair, tap) -> this chunk's partial map
  float* red = reinterpret_cast<float*>(XT);             // [wave][16][64]
  const long coci = static_cast<long>(Cout) * Cin;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    float* dst = part + (static_cast<long>(chunk) * 27 + kd * 9 + t) * coci;
    for (int e = tid; e < pairs * 1024; e += 512) {
      const int p = e >> 10, el = e & 1023, r = el >> 6, l = el & 63;
      const int pc = p / tiles_ci, pi = p - pc * tiles_ci;
      const int co = pc * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), ci = pi * 32 + (l & 31);
      float v = 0.f;
      for (int k = 0; k < ksplit; ++k) v += red[(p * ksplit + k) * 1024 + el];
      if (co < Cout && ci < Cin) dst[static_cast<long>(co) * Cin + ci] = v;
    }
    __syncthreads();
  }
}
