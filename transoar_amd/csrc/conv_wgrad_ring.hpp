// Weight gradient of a 3x3x3 / pad 1 convolution (stride 1 or 2) with up to 64 channels on either side over 10^6..10^7
// voxels: the 24 -> 48 / stride 2 and 48 -> 48 layers of stage 1 (backbones/encoder_blocks.py:28-51).
//   part[chunk][tap][co][ci] = sum over the chunk's dy voxels j:  dy[j][co] * x[S j + tap - 1][ci]
// conv3d_wgrad_kernel (the general voxel-major GEMM above) gives every (tap, channel tile) its own workgroups: x is
// fetched 27 times through L2 (4.2 GB for 48 -> 48 @ 80x80x128: 0.65 ms where the MFMAs take 0.14 and HBM 0.1).  Here a
// workgroup owns ONE filter plane kd and one (co tile, ci tile) of 32 x 32 channels, and walks W-rows of dy with h
// fastest, keeping the three x rows (kh) of its plane in an LDS ring -- x and dy pass through L2 3 (x 2 co / ci tiles)
// times instead of 27:
//   unit     64 consecutive dy voxels of a W-row (b, jd, jh, jw0..jw0+63)
//   GT       [2][64 voxels][64 B]                       the dy tile (32 channels), double-buffered
//   XT       [ring][columns][64 B]                      x rows ih = S jh + kh - 1, columns iw = S jw0 - 1 + c (66 for
//            S = 1; 129 for S = 2, stored de-interleaved by parity so that the 16 voxels of a K step are consecutive rows
//            for every kw); ring of 4 (S = 1: one new row per unit) or 6 (S = 2: two) -- the next unit's rows are
//            written while this unit's are read, one barrier per unit
//   4 waves  split the K axis: wave w multiplies voxels 16 w .. 16 w + 15 of the unit for all 9 taps (kh, kw) -- A and B
//            fragments by ds_read_b64_tr_b16 (64-byte pitch: 4 rows = all 64 banks), 9 accumulators of 32 x 32 fp32 per wave,
//            summed over the 4 waves once at the end.  Every wave runs the same code.
//   grid     chunks x variants, variant = (kd, co tile, ci tile) fastest: the variants of one spatial task run next to
//            each other on one XCD and share its L2; a persistent workgroup keeps its variant (the grid is a multiple of
//            the variant count) and its accumulators across all of its tasks.
// Output = conv3d_wgrad_kernel's partial maps: conv3d_wgrad_reduce_kernel finishes.  Included by conv_gemm.hip.
#pragma once

template <int S> struct RingGeom {
  static constexpr int kCols = 64 * S + 2;                       // x columns a unit touches (66 / 130, the last of S = 2 unused)
  static constexpr int kPlane = S == 1 ? 0 : 66;                 // S = 2: odd columns start at row kPlane of the staged x row
  static constexpr int kRows = S == 1 ? 68 : 132;                // staged rows (positions) per x row
  static constexpr int kRing = S == 1 ? 4 : 6;
  static constexpr int kNew = S;                                 // new x rows per unit
  __device__ static constexpr int position(int c) { return S == 1 ? c : (c & 1) * kPlane + (c >> 1); }
  __device__ static constexpr int tap_position(int kw) { return S == 1 ? kw : (kw == 1 ? kPlane : kw >> 1); }     // of voxel 0
};

template <int S>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_ring_kernel(
    const unsigned short* __restrict__ DY, const unsigned short* __restrict__ X, float* __restrict__ part, int N, int SD, int SH, int SW,
    int MD, int MH, int MW, int Cin, int Cout, int tiles_co, int tiles_ci, int h_chunks, int h_chunk) {
  typedef RingGeom<S> G;
  constexpr int kRowB = G::kRows * 64;                            // bytes of one staged x row
  constexpr int kXItems = (G::kNew * G::kCols * 4 + 255) / 256;   // 16-byte pieces of the new rows per thread
  __shared__ __attribute__((aligned(16))) unsigned char GT[2][64 * 64];
  __shared__ __attribute__((aligned(16))) unsigned char XT[G::kRing * kRowB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nv = 3 * tiles_co * tiles_ci;
  const int grid = static_cast<int>(gridDim.x);
  const int pos = (grid & 7) ? static_cast<int>(blockIdx.x) : static_cast<int>(blockIdx.x & 7) * (grid >> 3) + static_cast<int>(blockIdx.x >> 3);
  const int variant = pos % nv, chunk = pos / nv, chunks = grid / nv;
  const int kd = variant / (tiles_co * tiles_ci), co0 = ((variant / tiles_ci) % tiles_co) * 32, ci0 = (variant % tiles_ci) * 32;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int i = tid; i < 2 * 64 * 64 / 4; i += 256) reinterpret_cast<unsigned*>(&GT[0][0])[i] = 0u;      // channels beyond Cout / Cin stay zero
  for (int i = tid; i < G::kRing * kRowB / 4; i += 256) reinterpret_cast<unsigned*>(XT)[i] = 0u;

  // per-thread staging items: the same for every unit
  const int g_piece = tid & 3, g_vox = tid >> 2;
  const bool g_valid = co0 + g_piece * 8 < Cout;
  const unsigned g_lds = static_cast<unsigned>(g_vox * 64 + g_piece * 16);
  const unsigned g_goff = static_cast<unsigned>(g_vox * Cout + co0 + g_piece * 8);
  unsigned x_lds[kXItems], x_goff[kXItems];
  int x_meta[kXItems];                                 // new-row index | column << 8, or -1
  const long x_bias = static_cast<long>(Cin);          // column -1 of the first row: makes every item offset non-negative
#pragma unroll
  for (int k = 0; k < kXItems; ++k) {
    const int i = tid + 256 * k;
    const int piece = i & 3, q = i >> 2, nr = q / G::kCols, c = q - nr * G::kCols;
    const bool ok = nr < G::kNew && ci0 + piece * 8 < Cin && (S == 1 || c < 129);
    x_lds[k] = static_cast<unsigned>(G::position(c) * 64 + piece * 16);
    x_goff[k] = static_cast<unsigned>(x_bias + (static_cast<long>(nr) * SW + (c - 1)) * Cin + ci0 + piece * 8);
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
  const int tr_off = (16 * wave + 8 * kg + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
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
        for (int i = tid; i < 3 * G::kCols * 4; i += 256) {
          const int piece = i & 3, q = i >> 2, kh = q / G::kCols, c = q - kh * G::kCols;
          const int ih = S * jh + kh - 1, iw = S * jw0 + c - 1;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (ci0 + piece * 8 < Cin && static_cast<unsigned>(ih) < static_cast<unsigned>(SH) && static_cast<unsigned>(iw) < static_cast<unsigned>(SW) &&
              (S == 1 || c < 129))
            v = *reinterpret_cast<const u32x4*>(X + (((static_cast<long>(b) * SD + id) * SH + ih) * SW + iw) * Cin + ci0 + piece * 8);
          if (S == 1 || c < 129)
            *reinterpret_cast<u32x4*>(XT + ((ih + G::kRing) % G::kRing) * kRowB + G::position(c) * 64 + piece * 16) = v;
        }
        __syncthreads();
        if (jh + 1 < h_end) fetch(b, jd, id, jw0, jh + 1);
      }
      // unit jh + 1 (loaded one unit ago) -> the buffers nobody reads now; unit jh + 2 -> registers
      if (jh + 1 < h_end) stage(jh + 1);
      if (jh + 2 < h_end) fetch(b, jd, id, jw0, jh + 2);
      const s16x8 a = frag(GT[par] + tr_off);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const unsigned char* xrow = XT + ((S * jh + kh - 1 + G::kRing) % G::kRing) * kRowB + tr_off;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = mfma(a, frag(xrow + G::tap_position(kw) * 64), acc[kh * 3 + kw]);
      }
      __syncthreads();
    }
  }
  // ---- the four waves' accumulators -> one tile per tap -> this chunk's partial map
  float* red = reinterpret_cast<float*>(XT);             // [wave][16][64]
  const long coci = static_cast<long>(Cout) * Cin;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    float* dst = part + (static_cast<long>(chunk) * 27 + kd * 9 + t) * coci;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + 256 * j, r = e >> 6, l = e & 63;
      const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), ci = ci0 + (l & 31);
      if (co < Cout && ci < Cin) dst[static_cast<long>(co) * Cin + ci] = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
    }
    __syncthreads();
  }
}
