// Weight gradient of a 3x3x3 / stride 1 / pad 1 convolution with few channels (Cin, Cout <= 32) over 10^7 voxels --
// second version of conv3d_wgrad_lds.hpp (same unit walk, same partial layout):
//   dW[cout][cin][tap] = sum_v dy[v][cout] * x[v + tap][cin],   x, dy NDHWC bf16.
// The first version transposed both operands on their way INTO LDS with 2-byte writes (8 per 16 bytes loaded; 37 % of
// its LDS cycles were bank conflicts, profiles/r03_conv_pmc.txt), ran 3 waves per workgroup (1.5 waves per SIMD) and
// decomposed the unit index and every staging item by integer division per unit (209 VALU + 185 SALU instructions per
// unit and wave for 27 MFMAs).  Here the tiles are staged as they lie in memory ([voxel][channel], 64-byte pitch, 16-byte
// writes) and transposed on the way OUT by ds_read_b64_tr_b16 (4 rows x 64 bytes per 16 lanes = all 64 banks):
//   unit     64 consecutive voxels of a W-row (b, d, h, w0..w0+63), walked with h fastest
//   GT       [2][64 voxels][64 B]                 the dy tile, double-buffered
//   XT       [kd][ih mod 4][68 voxels][64 B]      the neighbouring x rows w0-1 .. w0+64, a 4-deep ring per kd: the
//            row the NEXT unit brings in (ih = h + 2) is written while this unit's three rows are being read, zero
//            outside the volume
//   4 waves  own taps 0-6 / 7-13 / 14-20 / 21-26 (tap = (kd*3 + kh)*3 + kw) with their 32 x 32 fp32 accumulators in
//            registers across all units of the workgroup; per 16-voxel K step a wave reads the dy fragment and one x
//            fragment per tap (2 transposing reads each; tap kw = the same read kw rows further down): 16 LDS reads
//            for 7 MFMAs and no VALU work on the operands.
//   pipeline per unit: [store unit u+1's rows (loaded one unit ago) into the free buffers] [issue unit u+2's loads]
//            [MFMAs of unit u] [ONE barrier].  Staging items and their offsets are per-thread constants, the unit
//            coordinates are wave-uniform.  (9 waves x 3 taps was tried: 1.5 ms, barrier-bound.)
//   tasks    (h chunk, W segment, (b, d) slice) with the slice fastest: see the note at the task loop
// Included by conv3d.hip (inside namespace transoar).
#pragma once

constexpr int kWtRows = 68;                    // staged x voxels per row: 66 used
constexpr int kWtThreads = 256;
constexpr int kWtSlot = kWtRows * 64;           // bytes of one staged x row

typedef short wt_s16x4 __attribute__((ext_vector_type(4)));

template <int T0, int NT>
__device__ __forceinline__ void wt_compute(f32x16 (&acc)[7], const unsigned char* GT, const unsigned char* XT, int h, int lane) {
  typedef __attribute__((address_space(3))) wt_s16x4 lds_s16x4;
  const int kg = lane >> 5;
  const int tr_off = (8 * kg + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  constexpr int R0 = T0 / 3, R1 = (T0 + NT - 1) / 3;             // (kd, kh) rows this wave touches
  const unsigned char* xrow[R1 - R0 + 1];
#pragma unroll
  for (int r = R0; r <= R1; ++r) xrow[r - R0] = XT + ((r / 3) * 4 + ((h + (r % 3) + 3) & 3)) * kWtSlot + tr_off;      // row ih = h + kh - 1
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const unsigned char* ap = GT + ks * 16 * 64 + tr_off;
    const u32x2c a0 = __builtin_bit_cast(u32x2c, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap)));
    const u32x2c a1 = __builtin_bit_cast(u32x2c, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + 4 * 64)));
    const u32x4c a = {a0[0], a0[1], a1[0], a1[1]};
#pragma unroll
    for (int t = T0; t < T0 + NT; ++t) {
      // tap kw of voxel v sits in tile column v + kw (column 0 is voxel w0 - 1): the fragment of tap kw is the same
      // transposing read kw rows further down.  (Cutting the kw = 1, 2 fragments out of one 12-voxel read with
      // v_alignbit / register moves cost 8 VALU instructions per row and K step: the kernel was VALU-bound.)
      const unsigned char* bp = xrow[t / 3 - R0] + (ks * 16 + t % 3) * 64;
      const u32x2c b0 = __builtin_bit_cast(u32x2c, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(bp)));
      const u32x2c b1 = __builtin_bit_cast(u32x2c, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(bp + 4 * 64)));
      const u32x4c b = {b0[0], b0[1], b1[0], b1[1]};
      acc[t - T0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[t - T0], 0, 0, 0);
    }
  }
}

template <int T0, int NT>
__device__ __forceinline__ void wt_store(const f32x16 (&acc)[7], float* out, int lane) {
  // partial[wg][tap][cout][cin]: D[row = cout (r & 3) + 8 (r >> 2) + 4 kg][col = cin]
  const int col = lane & 31, kg = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(T0 + t) * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + col] = acc[t][r];
}

template <int CIN8, int COUT8>
__global__ __launch_bounds__(kWtThreads, 2) void conv3d_k3_wgrad_tr(
    const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy, float* __restrict__ partial,
    int N, int D, int H, int W, int h_chunks, int h_chunk, int cin_total, int ci0, int cout_total, int co0) {
  constexpr int kXItems = (3 * 66 * CIN8 + kWtThreads - 1) / kWtThreads;       // one new h-row per kd
  __shared__ __attribute__((aligned(16))) unsigned char GT[2][64 * 64];
  __shared__ __attribute__((aligned(16))) unsigned char XT[12 * kWtSlot];      // [kd][ih mod 4][voxel][64 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int segs = W / 64;

  f32x16 acc[7];
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // channels beyond Cin / Cout and the voxels 66, 67 of every row stay zero for the whole kernel
  for (int i = tid; i < 2 * 64 * 64 / 4; i += kWtThreads) reinterpret_cast<unsigned int*>(&GT[0][0])[i] = 0u;
  for (int i = tid; i < 12 * kWtSlot / 4; i += kWtThreads) reinterpret_cast<unsigned int*>(XT)[i] = 0u;

  // per-thread staging items: the same for every unit
  u32x4c graw, xraw[kXItems];
  unsigned x_lds[kXItems], x_goff[kXItems];
  int x_meta[kXItems];                                 // kdd | vw << 8, or -1
  const long x_bias = (static_cast<long>(H) * W + 1) * cin_total;          // makes every item offset non-negative
#pragma unroll
  for (int k = 0; k < kXItems; ++k) {
    const int i = tid + k * kWtThreads;
    const int c8 = i % CIN8, q = i / CIN8, kdd = q / 66, vw = q % 66;
    x_lds[k] = static_cast<unsigned>(kdd * 4 * kWtSlot + vw * 64 + c8 * 16);
    x_goff[k] = static_cast<unsigned>(x_bias + (static_cast<long>(kdd - 1) * H * W + (vw - 1)) * cin_total + ci0 + c8 * 8);
    x_meta[k] = i < 3 * 66 * CIN8 ? (kdd | (vw << 8)) : -1;
  }
  const bool g_valid = tid < 64 * COUT8;
  const unsigned g_lds = static_cast<unsigned>((tid / COUT8) * 64 + (tid % COUT8) * 16);
  const unsigned g_goff = static_cast<unsigned>((tid / COUT8) * cout_total + co0 + (tid % COUT8) * 8);

  auto x_item = [&](long bd, int d, int w0, int kdd, int ih, int vw, int c8) -> u32x4c {
    const int id = d + kdd - 1, iw = w0 + vw - 1;
    u32x4c v = {0u, 0u, 0u, 0u};
    if (static_cast<unsigned>(id) < static_cast<unsigned>(D) && static_cast<unsigned>(ih) < static_cast<unsigned>(H) &&
        static_cast<unsigned>(iw) < static_cast<unsigned>(W))
      v = *reinterpret_cast<const u32x4c*>(x + (((bd - d + id) * H + ih) * W + iw) * cin_total + ci0 + c8 * 8);
    return v;
  };
  // dy tile of unit (bd, h, w0) and its new x rows ih = h + 1 -> registers
  auto fetch = [&](long bd, int d, int w0, int h) {
    const unsigned short* gb = dy + ((bd * H + h) * W + w0) * cout_total;
    graw = u32x4c{0u, 0u, 0u, 0u};
    if (g_valid) graw = *reinterpret_cast<const u32x4c*>(gb + g_goff);
    const unsigned short* xb = x + ((bd * H + h + 1) * W + w0) * cin_total - x_bias;
    const bool ih_ok = h + 1 < H;
#pragma unroll
    for (int k = 0; k < kXItems; ++k) {
      const int kdd = x_meta[k] & 0xff, vw = x_meta[k] >> 8;
      const bool ok = x_meta[k] >= 0 && ih_ok && static_cast<unsigned>(d + kdd - 1) < static_cast<unsigned>(D) &&
                      static_cast<unsigned>(w0 + vw - 1) < static_cast<unsigned>(W);
      xraw[k] = u32x4c{0u, 0u, 0u, 0u};
      if (ok) xraw[k] = *reinterpret_cast<const u32x4c*>(xb + x_goff[k]);
    }
  };
  // registers (unit with row h) -> the buffers that unit will read: GT[h & 1], ring slot of ih = h + 1
  auto stage = [&](int h) {
    if (g_valid) *reinterpret_cast<u32x4c*>(&GT[h & 1][g_lds]) = graw;
    const unsigned slot = static_cast<unsigned>(((h + 1) & 3) * kWtSlot);
#pragma unroll
    for (int k = 0; k < kXItems; ++k)
      if (x_meta[k] >= 0) *reinterpret_cast<u32x4c*>(XT + x_lds[k] + slot) = xraw[k];
  };

  // Tasks = (h chunk, W segment, (b, d) slice), the slice FASTEST, dealt round-robin to the persistent workgroups with
  // every XCD taking a contiguous range: the workgroups resident at the same time work on the same rows (h, segment) of
  // neighbouring d slices, so that the x rows a unit needs for kd = 0, 1, 2 -- read by three different workgroups --
  // come from L2 / the memory-side cache twice out of three times.  (With each workgroup walking its own contiguous
  // range of units, x came from HBM three times: 2.5 GB per call where dy + x are 1.26 GB.)
  const int nbd = N * D;
  const long n_tasks = static_cast<long>(nbd) * segs * h_chunks;
  const int grid = static_cast<int>(gridDim.x);
  const int first = (grid & 7) ? static_cast<int>(blockIdx.x) : static_cast<int>(blockIdx.x & 7) * (grid >> 3) + static_cast<int>(blockIdx.x >> 3);
  __syncthreads();
  for (long task = first; task < n_tasks; task += grid) {
    const long bd = task % nbd;
    const int r_ = static_cast<int>(task / nbd);
    const int w0 = (r_ % segs) * 64, hh = r_ / segs;
    const int d = static_cast<int>(bd % D);
    const int h_beg = hh * h_chunk, h_end = min(H, h_beg + h_chunk);
    for (int h = h_beg; h < h_end; ++h) {
      const int par = h & 1;
      if (h == h_beg) {
        // first unit of a task: its dy tile and all 9 rows straight through (nothing was staged ahead; the barrier at
        // the end of the previous unit covers the buffers)
        if (g_valid) *reinterpret_cast<u32x4c*>(&GT[par][g_lds]) =
            *reinterpret_cast<const u32x4c*>(dy + ((bd * H + h) * W + w0) * cout_total + g_goff);
        for (int i = tid; i < 9 * 66 * CIN8; i += kWtThreads) {
          const int c8 = i % CIN8, q = i / CIN8;
          const int vw = q % 66, r = q / 66;                          // r = kd*3 + kh
          const int ih = h + r % 3 - 1;
          *reinterpret_cast<u32x4c*>(XT + ((r / 3) * 4 + ((ih + 4) & 3)) * kWtSlot + vw * 64 + c8 * 16) = x_item(bd, d, w0, r / 3, ih, vw, c8);
        }
        __syncthreads();
        if (h + 1 < h_end) fetch(bd, d, w0, h + 1);
      }
      // unit h + 1 (loaded one unit ago) -> the buffers nobody reads now; unit h + 2 -> registers
      if (h + 1 < h_end) stage(h + 1);
      if (h + 2 < h_end) fetch(bd, d, w0, h + 2);
      switch (wave) {
        case 0: wt_compute<0, 7>(acc, GT[par], XT, h, lane); break;
        case 1: wt_compute<7, 7>(acc, GT[par], XT, h, lane); break;
        case 2: wt_compute<14, 7>(acc, GT[par], XT, h, lane); break;
        default: wt_compute<21, 6>(acc, GT[par], XT, h, lane); break;
      }
      __syncthreads();
    }
  }
  float* out = partial + static_cast<long>(blockIdx.x) * 27 * 1024;
  switch (wave) {
    case 0: wt_store<0, 7>(acc, out, lane); break;
    case 1: wt_store<7, 7>(acc, out, lane); break;
    case 2: wt_store<14, 7>(acc, out, lane); break;
    default: wt_store<21, 6>(acc, out, lane); break;
  }
}
