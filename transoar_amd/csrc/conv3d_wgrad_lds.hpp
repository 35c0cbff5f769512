// Weight gradient of a 3x3x3 / stride 1 / pad 1 convolution with few channels (Cin, Cout <= 32),
// the contraction running over 10^7 voxels:   dW[cout][cin][tap] = sum_v dy[v][cout] * x[v + tap][cin]
// with x, dy NDHWC bf16.  MFMA wants 8 consecutive contraction elements (voxels) per lane, memory has
// the channels contiguous instead -> both operands are transposed on their way into LDS:
//   unit of work   64 consecutive voxels of a W-row (b, d, h, w0..w0+63); units are walked with h fastest
//   GT[cout][64]                      dy tile, voxel-contiguous
//   XT[kd][ih mod 3][cin][66(+pad)]   the 9 neighbouring rows of x, w0-1 .. w0+64, zero outside the volume;
//                                     a 3-deep ring per kd: a step in h brings in one new row per kd
//   3 waves, wave kd owns the 9 taps (kd, kh, kw): per 16-voxel k-step one A fragment (ds_read_b128 of
//   GT) and per (kd,kh) row two aligned ds_read_b128 of XT from which the three kw-shifted B fragments
//   are cut with v_alignbit; 9 x v_mfma_f32_32x32x16_bf16.  Accumulators (9 x 32x32 fp32 per wave) live
//   in registers across all units of the workgroup; the caller sums the per-workgroup partials.
// Included by conv3d.hip (inside namespace transoar).
#pragma once

constexpr int kWgGPitch = 72;                  // elements per GT row (64 + pad against bank conflicts)
constexpr int kWgXPitch = 88;                  // elements per XT row: 66 used, 16-byte aligned windows up to 80
constexpr int kWgThreads = 192;

__device__ __forceinline__ unsigned int wg_alignbit(unsigned int hi, unsigned int lo, int bits) {
  return bits == 0 ? lo : static_cast<unsigned int>(((static_cast<unsigned long long>(hi) << 32) | lo) >> bits);
}

template <int CIN8, int COUT8>
__global__ __launch_bounds__(kWgThreads) void conv3d_k3_wgrad_lds(
    const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy, float* __restrict__ partial,
    int N, int D, int H, int W, long n_units, int units_per_wg, int cin_total, int ci0, int cout_total, int co0) {
  // this launch: channels [ci0, ci0 + 8*CIN8) of x rows that are cin_total wide, same for dy
  constexpr int kGItems = (64 * COUT8 + kWgThreads - 1) / kWgThreads;
  constexpr int kXItems = (3 * 66 * CIN8 + kWgThreads - 1) / kWgThreads;      // one new h-row per kd
  __shared__ __attribute__((aligned(16))) unsigned short GT[32 * kWgGPitch];
  __shared__ __attribute__((aligned(16))) unsigned short XT[9 * 32 * kWgXPitch];   // [kd][ih mod 3][cin][w]
  const int tid = threadIdx.x, lane = tid & 63, kd = tid >> 6;       // wave = kd
  const int col = lane & 31, kg = lane >> 5;
  const int segs = W / 64;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // rows of GT / XT beyond Cout / Cin stay zero for the whole kernel
  for (int i = tid; i < 32 * kWgGPitch / 2; i += kWgThreads) reinterpret_cast<unsigned int*>(GT)[i] = 0u;
  for (int i = tid; i < 9 * 32 * kWgXPitch / 2; i += kWgThreads) reinterpret_cast<unsigned int*>(XT)[i] = 0u;

  // Units are walked with h fastest: unit u = (outer, h), outer = (b*D + d)*segs + seg.  Going from h to
  // h+1 only the x rows ih = h+2 are new (one per kd): they replace the oldest slot of a 3-deep ring per
  // kd.  Their 16-byte loads and the dy tile's are issued one unit ahead (in flight during the MFMAs).
  u32x4c graw[kGItems], xraw[kXItems];
  auto x_item = [&](long bd, int d, int w0, int kdd, int ih, int vw, int c8) -> u32x4c {
    const int id = d + kdd - 1, iw = w0 + vw - 1;
    u32x4c v = {0u, 0u, 0u, 0u};
    if (static_cast<unsigned>(id) < static_cast<unsigned>(D) && static_cast<unsigned>(ih) < static_cast<unsigned>(H) &&
        static_cast<unsigned>(iw) < static_cast<unsigned>(W))
      v = *reinterpret_cast<const u32x4c*>(x + (((bd - d + id) * H + ih) * W + iw) * cin_total + ci0 + c8 * 8);
    return v;
  };
  auto x_store = [&](int kdd, int ih, int vw, int c8, const u32x4c& v) {
    unsigned short* dst = XT + ((kdd * 3 + (ih + 3) % 3) * 32 + c8 * 8) * kWgXPitch + vw;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dst[(2 * e) * kWgXPitch] = static_cast<unsigned short>(v[e] & 0xffffu);
      dst[(2 * e + 1) * kWgXPitch] = static_cast<unsigned short>(v[e] >> 16);
    }
  };
  auto fetch = [&](long u) {                  // dy tile of unit u, and (h > 0) its new x rows ih = h + 1
    const int h = static_cast<int>(u % H);
    const long outer = u / H;
    const int w0 = static_cast<int>(outer % segs) * 64;
    const long bd = outer / segs;
    const int d = static_cast<int>(bd % D);
#pragma unroll
    for (int k = 0; k < kGItems; ++k) {
      const int i = tid + k * kWgThreads;
      graw[k] = u32x4c{0u, 0u, 0u, 0u};
      if (i < 64 * COUT8) {
        const int c8 = i / 64, v = i - c8 * 64;
        graw[k] = *reinterpret_cast<const u32x4c*>(dy + ((bd * H + h) * W + w0 + v) * cout_total + co0 + c8 * 8);
      }
    }
    if (h > 0) {
#pragma unroll
      for (int k = 0; k < kXItems; ++k) {
        const int i = tid + k * kWgThreads;
        xraw[k] = u32x4c{0u, 0u, 0u, 0u};
        if (i < 3 * 66 * CIN8) {
          const int vw = i % 66, t = i / 66;
          xraw[k] = x_item(bd, d, w0, t / CIN8, h + 1, vw, t % CIN8);
        }
      }
    }
  };

  const long u0 = static_cast<long>(blockIdx.x) * units_per_wg;
  const long u_end = min(u0 + units_per_wg, n_units);
  if (u0 < u_end) fetch(u0);
  __syncthreads();
  for (long u = u0; u < u_end; ++u) {
    const int h = static_cast<int>(u % H);
    // ---- registers -> LDS, transposed (2-byte writes)
#pragma unroll
    for (int k = 0; k < kGItems; ++k) {
      const int i = tid + k * kWgThreads;
      if (i < 64 * COUT8) {
        const int c8 = i / 64, v = i - c8 * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          GT[(c8 * 8 + 2 * e) * kWgGPitch + v] = static_cast<unsigned short>(graw[k][e] & 0xffffu);
          GT[(c8 * 8 + 2 * e + 1) * kWgGPitch + v] = static_cast<unsigned short>(graw[k][e] >> 16);
        }
      }
    }
    if (h > 0 && u > u0) {
#pragma unroll
      for (int k = 0; k < kXItems; ++k) {
        const int i = tid + k * kWgThreads;
        if (i < 3 * 66 * CIN8) {
          const int vw = i % 66, t = i / 66;
          x_store(t / CIN8, h + 1, vw, t % CIN8, xraw[k]);
        }
      }
    } else {
      // first unit of the workgroup or of an (b, d, segment) column: all 9 rows, straight through
      const long outer = u / H;
      const int w0 = static_cast<int>(outer % segs) * 64;
      const long bd = outer / segs;
      const int d = static_cast<int>(bd % D);
      for (int i = tid; i < 9 * 66 * CIN8; i += kWgThreads) {
        const int vw = i % 66, t = i / 66;
        const int c8 = t % CIN8, r = t / CIN8;                    // r = kd*3 + kh
        const int ih = h + r % 3 - 1;
        x_store(r / 3, ih, vw, c8, x_item(bd, d, w0, r / 3, ih, vw, c8));
      }
    }
    __syncthreads();
    if (u + 1 < u_end) fetch(u + 1);          // in flight during the MFMA phase

    // ---- 4 k-steps of 16 voxels; this wave: taps (kd, kh, kw)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int v0 = ks * 16 + 8 * kg;                            // first voxel of this lane's 8
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(GT + col * kWgGPitch + v0);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        // XT column index of voxel v, tap kw:  v + kw  (column 0 is w0-1); row slot of ih = h + kh - 1
        const unsigned short* src = XT + ((kd * 3 + (h + kh + 2) % 3) * 32 + col) * kWgXPitch + v0;
        const u32x4c lo = *reinterpret_cast<const u32x4c*>(src);
        const u32x4c hi = *reinterpret_cast<const u32x4c*>(src + 8);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          u32x4c b;
          if (kw == 0) {
            b = lo;
          } else if (kw == 1) {
            b[0] = wg_alignbit(lo[1], lo[0], 16); b[1] = wg_alignbit(lo[2], lo[1], 16);
            b[2] = wg_alignbit(lo[3], lo[2], 16); b[3] = wg_alignbit(hi[0], lo[3], 16);
          } else {      // shift by 32 bits: whole dwords
            b[0] = lo[1]; b[1] = lo[2]; b[2] = lo[3]; b[3] = hi[0];
          }
          acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b), acc[kh * 3 + kw], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // partial[wg][tap = (kd*3+kh)*3+kw][cout][cin]
  float* out = partial + (static_cast<long>(blockIdx.x) * 27 + kd * 9) * 1024;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[t * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + col] = acc[t][r];
}
