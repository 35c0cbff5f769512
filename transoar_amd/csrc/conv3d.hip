// 3x3x3 convolution (pad 1, stride 1 or 2) of the AttnFPN backbone on gfx950:
// implicit GEMM on the bf16 matrix cores, fp32 accumulation, channels-last.
//
// Replaces the cuDNN calls behind nn.Conv3d in the reference's encoder stages
// (transoar/models/backbones/encoder_blocks.py:28-48) and FPN output convs
// (transoar/models/backbones/attn_fpn.py:65-73,126).  Three kernels:
//
//   conv3d_k3_igemm   y = conv(x, w) (+bias).  Also computes dgrad: for a
//                     stride-1 layer dx = conv(dy, flipped/transposed w); for a
//                     stride-2 layer the same over the zero-dilated dy (DIL).
//                     GEMM view: D[cout][voxel] = sum_K W[cout][K] X[K][voxel],
//                     K = (tap, cin) walked in chunks of 8 input channels.
//   conv3d_k3_wgrad   dW[cout][(tap,cin)] = sum_voxels dy[voxel][cout] *
//                     x[voxel+tap][cin]; here K = voxels, so the operands come
//                     from channels-FIRST copies (W-contiguous), the x copy
//                     pre-shifted along W for the three kw taps.
//   conv3d_c1_fwd     Cin == 1 first layer: a stencil, HBM-bound, no MFMA.
//
// No LDS tiling yet: every MFMA operand fragment is one 16-byte buffer load per
// lane (8 bf16 along the contraction axis, exactly the 32x32x16 fragment), out
// of range taps / channels are handled by the buffer's hardware bounds check
// (offset -> 0xfffffff0 -> zeros), and each wave register-blocks MT x NT tiles
// of 32x32 so a fragment is reused MT or NT times.  D is computed transposed
// (rows = cout) so that a lane ends up with 4 consecutive output channels of
// one voxel per accumulator quad -> 8-byte stores into the NDHWC output.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <stdint.h>

#include "../../include/transoar_conv3d.h"

namespace transoar {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4c = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2c = __attribute__((ext_vector_type(2))) unsigned int;

__device__ __forceinline__ unsigned short f2bf(float f) {
  return __builtin_bit_cast(unsigned short, static_cast<__bf16>(f));      // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
}
// (bf16(b) << 16) | bf16(a), round to nearest even: one v_cvt_pk_bf16_f32 (the integer sequence of f2bf is 7 VALU
// instructions per value; conv3d_k3_lds spent a fifth of its 7 800 VALU instructions per wave there)
__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
__device__ __forceinline__ float bf2f(unsigned short b) {
  return __uint_as_float(static_cast<unsigned int>(b) << 16);
}

constexpr unsigned kOOB = 0xfffffff0u;

struct ConvGeom {
  int N, D, H, W, Cin;        // input (for DIL: the un-dilated dy grid)
  int Do, Ho, Wo, Cout;       // output
  int CinP;                   // Cin rounded up to 8 (row length of the packed weights)
  int stride;                 // 1 or 2 (always 1 when DIL)
};

// ---------------------------------------------------------------------------
// implicit GEMM, K = (tap, cin)
//   x   (N, D, H, W, Cin) bf16          wk (27, Cout, CinP) bf16
//   y   (N, Do, Ho, Wo, Cout) bf16      bias (Cout) fp32 or null
// DIL: input is the zero-dilated (x2) grid of x; coordinate u maps to u/2 when
// even (transposed convolution of a stride-2 layer).
// ---------------------------------------------------------------------------
template <int MT, int NT, bool DIL>
__global__ __launch_bounds__(256) void conv3d_k3_igemm(const unsigned short* __restrict__ x,
                                                       const unsigned short* __restrict__ wk,
                                                       const float* __restrict__ bias,
                                                       unsigned short* __restrict__ y, ConvGeom g,
                                                       long n_vox, unsigned x_bytes, unsigned w_bytes) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col = lane & 31;       // voxel column inside an M tile / cout row inside an N tile
  const int half = lane >> 5;      // which 8-wide K chunk of the 16-wide MFMA step
  const long m_base = (static_cast<long>(blockIdx.x) * 4 + wave) * (32 * MT);
  if (m_base >= n_vox) return;
  const int n_base = blockIdx.y * (32 * NT);

  const __amdgpu_buffer_rsrc_t xr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(x), 0, static_cast<int>(x_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t wr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wk), 0, static_cast<int>(w_bytes), 0x00020000);

  // per M tile: this lane's output voxel -> linear index of input tap (0,0,0) and validity bits
  int base_lin[MT];      // !DIL: ((n*D + id0)*H + ih0)*W + iw0   (may be negative)
  int okbits[MT];        // bit kd | bit 3+kh | bit 6+kw : that tap coordinate is in range
  int sd[MT][3], sh[MT][3], sw[MT][3];   // DIL only: source coordinates per tap index (or -1)
  int nD[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const long v = m_base + t * 32 + col;
    const bool live = v < n_vox;
    const long vv = live ? v : 0;
    const int ow = static_cast<int>(vv % g.Wo);
    const long r1 = vv / g.Wo;
    const int oh = static_cast<int>(r1 % g.Ho);
    const long r2 = r1 / g.Ho;
    const int od = static_cast<int>(r2 % g.Do);
    const int n = static_cast<int>(r2 / g.Do);
    const int id0 = od * g.stride - 1, ih0 = oh * g.stride - 1, iw0 = ow * g.stride - 1;
    int bits = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (DIL) {
        const int ud = id0 + k, uh = ih0 + k, uw = iw0 + k;
        sd[t][k] = (ud >= 0 && !(ud & 1) && (ud >> 1) < g.D) ? (ud >> 1) : -1;
        sh[t][k] = (uh >= 0 && !(uh & 1) && (uh >> 1) < g.H) ? (uh >> 1) : -1;
        sw[t][k] = (uw >= 0 && !(uw & 1) && (uw >> 1) < g.W) ? (uw >> 1) : -1;
      } else {
        bits |= (static_cast<unsigned>(id0 + k) < static_cast<unsigned>(g.D) ? 1 : 0) << k;
        bits |= (static_cast<unsigned>(ih0 + k) < static_cast<unsigned>(g.H) ? 1 : 0) << (3 + k);
        bits |= (static_cast<unsigned>(iw0 + k) < static_cast<unsigned>(g.W) ? 1 : 0) << (6 + k);
      }
    }
    okbits[t] = live ? bits : 0;
    if (DIL && !live) {
#pragma unroll
      for (int k = 0; k < 3; ++k) sd[t][k] = -1;
    }
    nD[t] = n * g.D;
    base_lin[t] = ((n * g.D + id0) * g.H + ih0) * g.W + iw0;
  }

  f32x16 acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;

  const int cpt = g.CinP >> 3;            // 8-channel chunks per tap
  const int n_chunks = 27 * cpt;
  const int HW = g.H * g.W;
  const unsigned cin_bytes = static_cast<unsigned>(g.Cin) * 2u;

  // this lane-half's chunk: q = 2*step + half -> (tap = kd,kh,kw ; c8)
  int c8 = half, kd = 0, kh = 0, kw = 0, tap = 0;
  while (c8 >= cpt) { c8 -= cpt; ++tap; if (++kw == 3) { kw = 0; if (++kh == 3) { kh = 0; ++kd; } } }

  for (int q = half; q < n_chunks + half; q += 2) {
    const bool chunk_ok = q < n_chunks && c8 * 8 < g.Cin;
    // weight fragments: A[i = cout][k] = wk[tap][cout][c8*8 .. +8]
    bf16x8 wf[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const int co = n_base + a * 32 + col;
      const unsigned off = (co < g.Cout && q < n_chunks)
                               ? (static_cast<unsigned>(tap * g.Cout + co) * g.CinP + c8 * 8) * 2u
                               : kOOB;
      wf[a] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, 0));
    }
    // activation fragments: B[k][j = voxel] = x[voxel @ tap][c8*8 .. +8]
    bf16x8 xf[MT];
    const int delta = kd * HW + kh * g.W + kw;
    const int need = (1 << kd) | (8 << kh) | (64 << kw);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      unsigned off;
      if (DIL) {
        const int d = kd == 0 ? sd[t][0] : (kd == 1 ? sd[t][1] : sd[t][2]);
        const int h = kh == 0 ? sh[t][0] : (kh == 1 ? sh[t][1] : sh[t][2]);
        const int w = kw == 0 ? sw[t][0] : (kw == 1 ? sw[t][1] : sw[t][2]);
        const bool ok = chunk_ok && (d | h | w) >= 0;
        const int lin = ((nD[t] + d) * g.H + h) * g.W + w;
        off = ok ? static_cast<unsigned>(lin) * cin_bytes + c8 * 16 : kOOB;
      } else {
        const bool ok = chunk_ok && (okbits[t] & need) == need;
        off = ok ? static_cast<unsigned>(base_lin[t] + delta) * cin_bytes + c8 * 16 : kOOB;
      }
      xf[t] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
    }
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int t = 0; t < MT; ++t)
        acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[a], xf[t], acc[a][t], 0, 0, 0);
    // advance this half's chunk by 2
    c8 += 2;
    while (c8 >= cpt) { c8 -= cpt; ++tap; if (++kw == 3) { kw = 0; if (++kh == 3) { kh = 0; ++kd; } } }
  }

  // epilogue: D[i = cout][j = voxel]; lane: col j, rows (r&3) + 8*(r>>2) + 4*half
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const long v = m_base + t * 32 + col;
    if (v >= n_vox) continue;
    unsigned short* yrow = y + v * g.Cout;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int co = n_base + a * 32 + 8 * gq + 4 * half;
        if (co < g.Cout) {      // Cout % 4 == 0 (checked on the host)
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[a][t][4 * gq + e] + (bias ? bias[co + e] : 0.f);
          u32x2c pk;
          pk[0] = pack2_bf16(o[0], o[1]);
          pk[1] = pack2_bf16(o[2], o[3]);
          *reinterpret_cast<u32x2c*>(yrow + co) = pk;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// implicit GEMM with the input halo tile in LDS (stride 1, Cin = 8 * CPT <= 24, Cout <= 32: the full-resolution
// layers -- stem, 24 -> 24 forward and its data gradient).  conv3d_k3_igemm fetches every activation fragment
// from global memory once per tap (27 x, 5 buffer loads per 4 MFMAs: bound by the vector-memory pipe at
// ~20 % of the matrix cores); here a workgroup owns 4 x 4 x 32 output voxels at a time, stages their 6 x 6 x 34 halo
// once (2.4 x the tile instead of 27 x) and the fragments are 16-byte LDS reads.  With the voxel pitch of
// 16 / 48 bytes the 16 lanes of a quarter wave start in 16 distinct 4-bank groups: conflict-free.
// Wave w owns the four 32-voxel rows of depth w; D is computed transposed as above.
// ---------------------------------------------------------------------------
template <int I> struct ConvIntC { static constexpr int value = I; };
template <int I, int N, typename F>
__device__ __forceinline__ void static_for_conv(F&& f) {
  if constexpr (I < N) {
    f(ConvIntC<I>{});
    static_for_conv<I + 1, N>(f);
  }
}

constexpr int kLtD = 4, kLtH = 4, kLtW = 32;
constexpr int kLtHaloVox = (kLtD + 2) * (kLtH + 2) * (kLtW + 2);

// STATS (round 4): the per-channel sum and sum of squares of the (bf16-rounded) outputs -- what the InstanceNorm that
// follows needs -- are taken from the accumulators in the epilogue instead of by a second pass over the 629-MB map:
// per tile a lane sums its 4 rows, a reduce-scatter over the 32 lanes of a wave half (one channel quad at a time; xor 16: sums | squares,
// xor 8 and 4: the quad's channels, two butterfly steps) leaves one running value per quad and lane; at the end of the workgroup's row of
// tiles they go through LDS into stat_part[row of tiles][sum | sumsq][32 channels] (fp32, <= 4 096 voxels each; a second
// kernel adds the rows in fp64: instnorm.hip).
template <int CPT, bool C1 = false, bool STATS = false>      // C1: the input has ONE channel (2 bytes per voxel), zero-extended to 8 while staging
__global__ __launch_bounds__(256, 2) void conv3d_k3_lds(const unsigned short* __restrict__ x,
                                                        const unsigned short* __restrict__ wk,
                                                        const float* __restrict__ bias,
                                                        unsigned short* __restrict__ y, ConvGeom g, int tiles_d,
                                                        int tiles_h, int tiles_w, long n_tiles, unsigned x_bytes,
                                                        unsigned w_bytes, float* __restrict__ stat_part = nullptr) {
  constexpr int VP = CPT * 16;                    // bytes per voxel (Cin = 8 * CPT channels)
  constexpr int MT = kLtH;
  __shared__ __attribute__((aligned(16))) unsigned char halo[kLtHaloVox * VP];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;

  // A workgroup walks the tiles_w tiles of one (n, td, th) row; rows in contiguous ranges per XCD (workgroups are
  // dealt round-robin to the 8 XCDs): neighbouring rows share halo voxels through the same L2.  The halo of the
  // next tile is fetched into registers while this one is multiplied.
  const long per = (n_tiles + 7) >> 3;            // n_tiles: rows here
  const long trow = static_cast<long>(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (trow >= n_tiles || (blockIdx.x >> 3) >= per) return;
  const int th = static_cast<int>(trow % tiles_h);
  const long t2 = trow / tiles_h;
  const int td = static_cast<int>(t2 % tiles_d);
  const int n = static_cast<int>(t2 / tiles_d);
  const int d0 = td * kLtD, h0 = th * kLtH;

  const __amdgpu_buffer_rsrc_t xr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(x), 0, static_cast<int>(x_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t wr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wk), 0, static_cast<int>(w_bytes), 0x00020000);

  // ---- halo -> registers -> LDS, 16-byte pieces; outside the volume: zeros (buffer bounds check)
  constexpr int PIECES = kLtHaloVox * CPT;
  constexpr int ROUNDS = (PIECES + 255) / 256;
  u32x4c stage[ROUNDS];
  auto fetch = [&](int w0) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int p = r * 256 + static_cast<int>(threadIdx.x);
      const int hv = p / CPT, c = p - hv * CPT;
      const int row = hv / (kLtW + 2), wi = hv - row * (kLtW + 2);
      const int hd = row / (kLtH + 2), hh = row - hd * (kLtH + 2);
      const int d = d0 - 1 + hd, h = h0 - 1 + hh, w = w0 - 1 + wi;
      const bool ok = p < PIECES && static_cast<unsigned>(d) < static_cast<unsigned>(g.D) &&
                      static_cast<unsigned>(h) < static_cast<unsigned>(g.H) && static_cast<unsigned>(w) < static_cast<unsigned>(g.W);
      if constexpr (C1) {
        const unsigned off = ok ? static_cast<unsigned>(((n * g.D + d) * g.H + h) * g.W + w) * 2u : kOOB;
        const unsigned short v = __builtin_amdgcn_raw_buffer_load_b16(xr, off, 0, 0);
        stage[r] = u32x4c{static_cast<unsigned>(v), 0u, 0u, 0u};
      } else {
        const unsigned off = ok ? (static_cast<unsigned>(((n * g.D + d) * g.H + h) * g.W + w) * CPT + c) * 16u : kOOB;
        stage[r] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
      }
    }
  };
  float srun[4] = {0.f, 0.f, 0.f, 0.f};     // STATS: this lane's running value per channel quad (see the epilogue)
  fetch(0);
  for (int tw = 0; tw < tiles_w; ++tw) {
  const int w0 = tw * kLtW;
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int p = r * 256 + static_cast<int>(threadIdx.x);
    if (p < PIECES) *reinterpret_cast<u32x4c*>(halo + p * 16) = stage[r];
  }
  __syncthreads();
  if (tw + 1 < tiles_w) fetch(w0 + kLtW);

  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // K = (kd | kh, kw, c8): a run-time loop over the three kd planes, the 9 * CPT chunks of a plane as
  // ceil(9 CPT / 2) fully unrolled K-steps (half-wave h takes chunk 2 * step + h; an odd last chunk multiplies by
  // zero weights).  Inside a plane (tap, c8) are compile-time constants per half: the LDS offset of a fragment is a
  // select between two immediates.  The weight fragment (global, L1) is fetched one step ahead.
  constexpr int n_chunks = 9 * CPT;
  constexpr int n_steps = (n_chunks + 1) / 2;
  const unsigned char* lane_base = halo + ((wave * (kLtH + 2)) * (kLtW + 2) + col) * VP;
  const unsigned w_lane = static_cast<unsigned>(col) * g.CinP * 2u;
  const bool co_ok = col < g.Cout;
  const unsigned w_tap = static_cast<unsigned>(g.Cout) * g.CinP * 2u;       // bytes per tap
  if constexpr (CPT == 3 && !C1) {
    // 24 channels (round 4).  The loop below reads one activation fragment per MFMA: 1 KiB of LDS per 32 matrix-core
    // cycles and wave, i.e. the CU's LDS pipe (128 B/clk) and its matrix cores saturate together and each ends up half
    // idle (0.70 ms on 24 -> 24 at 160 x 160 x 256 against 0.29 ms of MFMA time).  The fragment of output row t at
    // filter row kh is halo row t + kh: ordering K so that the three kh of one (kw, channel-chunk pair) are adjacent,
    // SIX row reads serve the TWELVE MFMAs of a group (3 kh x 4 output rows).  Per kd plane five groups:
    //   kw = 0, 1, 2:  half 0 takes channels 0-7, half 1 channels 8-15 of tap (kh, kw)
    //   group 3:       channels 16-23 of taps (kh, 0) | (kh, 1);   group 4: channels 16-23 of tap (kh, 2) | nothing
    // 15 K-steps per plane instead of 14 (+7 % MFMAs), half the LDS reads.  The three weight fragments of a group are
    // fetched a group ahead (global, L1-resident).
    auto load_w = [&](auto gc, int kd, bf16x8 (&wf)[3]) {
      constexpr int gi = decltype(gc)::value;
      constexpr int kw0 = gi < 3 ? gi : (gi == 3 ? 0 : 2), kw1 = gi < 3 ? gi : (gi == 3 ? 1 : 2);
      constexpr int c80 = gi < 3 ? 0 : 2, c81 = gi < 3 ? 1 : 2;
      const bool ok = co_ok && (half ? gi != 4 : true);
      const unsigned base = kd * 9 * w_tap + w_lane + (half ? kw1 * w_tap + c81 * 16 : kw0 * w_tap + c80 * 16);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
        wf[kh] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, ok ? base + kh * 3 * w_tap : kOOB, 0, 0));
    };
    bf16x8 wcur[3];
    load_w(ConvIntC<0>{}, 0, wcur);
#pragma unroll 1
    for (int kd = 0; kd < 3; ++kd) {
      const unsigned char* plane = lane_base + kd * (kLtH + 2) * (kLtW + 2) * VP;
      static_for_conv<0, 5>([&](auto gc) {
        constexpr int gi = decltype(gc)::value;
        constexpr int kw0 = gi < 3 ? gi : (gi == 3 ? 0 : 2), kw1 = gi < 3 ? gi : (gi == 3 ? 1 : 2);
        constexpr int c80 = gi < 3 ? 0 : 2, c81 = gi < 3 ? 1 : 2;
        bf16x8 wnext[3];
        if constexpr (gi + 1 < 5) load_w(ConvIntC<gi + 1>{}, kd, wnext);
        else load_w(ConvIntC<0>{}, kd + 1, wnext);             // (kd + 1 == 3: past the filter, reads zeros, unused)
        const unsigned char* src = plane + (half ? kw1 * VP + c81 * 16 : kw0 * VP + c80 * 16);
        bf16x8 xr[MT + 2];
#pragma unroll
        for (int r = 0; r < MT + 2; ++r) xr[r] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4c*>(src + r * (kLtW + 2) * VP));
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wcur[kh], xr[t + kh], acc[t], 0, 0, 0);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) wcur[kh] = wnext[kh];
      });
    }
  } else {
#pragma unroll 1
  for (int kd = 0; kd < 3; ++kd) {
    const unsigned char* plane = lane_base + kd * (kLtH + 2) * (kLtW + 2) * VP;
    const unsigned w_plane = kd * 9 * w_tap + w_lane;
    auto weight_fragment = [&](auto sc) {
      constexpr int step = decltype(sc)::value;
      constexpr int q0 = 2 * step, q1 = 2 * step + 1;
      constexpr int tap0 = q0 / CPT, c80 = q0 % CPT, tap1 = (q1 < n_chunks ? q1 : q0) / CPT, c81 = (q1 < n_chunks ? q1 : q0) % CPT;
      const int tap = half ? tap1 : tap0, c8 = half ? c81 : c80;
      const bool chunk_ok = half ? (q1 < n_chunks) : true;
      const unsigned woff = (co_ok && chunk_ok) ? w_plane + tap * w_tap + c8 * 16 : kOOB;
      return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, woff, 0, 0));
    };
    bf16x8 wf_cur = weight_fragment(ConvIntC<0>{});
    static_for_conv<0, n_steps>([&](auto sc) {
      constexpr int step = decltype(sc)::value;
      constexpr int q0 = 2 * step, q1 = 2 * step + 1;
      constexpr int tap0 = q0 / CPT, c80 = q0 % CPT, tap1 = (q1 < n_chunks ? q1 : q0) / CPT, c81 = (q1 < n_chunks ? q1 : q0) % CPT;
      constexpr int off0 = ((tap0 / 3) * (kLtW + 2) + tap0 % 3) * VP + c80 * 16;
      constexpr int off1 = ((tap1 / 3) * (kLtW + 2) + tap1 % 3) * VP + c81 * 16;
      bf16x8 wf_next = wf_cur;
      if constexpr (step + 1 < n_steps) wf_next = weight_fragment(ConvIntC<step + 1>{});
      // activation fragments: B[k][j = voxel] = halo[(wave + kd, t + kh, col + kw)][c8*8 .. +8]
      const unsigned char* src = plane + (half ? off1 : off0);
      bf16x8 xf[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) xf[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4c*>(src + t * (kLtW + 2) * VP));
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf_cur, xf[t], acc[t], 0, 0, 0);
      wf_cur = wf_next;
    });
  }
  }

  // epilogue: D[i = cout][j = voxel]; lane: col j, rows (r&3) + 8*(r>>2) + 4*half
  const int od = d0 + wave, ow = w0 + col;
  if (od < g.D && ow < g.W) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int oh = h0 + t;
      if (oh >= g.H) continue;
      unsigned short* yrow = y + (static_cast<long>((n * g.D + od) * g.H + oh) * g.W + ow) * g.Cout;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int co = 8 * gq + 4 * half;
        if (co < g.Cout) {      // Cout % 4 == 0 (checked on the host)
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[t][4 * gq + e] + (bias ? bias[co + e] : 0.f);
          u32x2c pk;
          pk[0] = pack2_bf16(o[0], o[1]);
          pk[1] = pack2_bf16(o[2], o[3]);
          *reinterpret_cast<u32x2c*>(yrow + co) = pk;
        }
      }
    }
  }
  if (STATS) {
    // One channel quad (gq) at a time -- eight temporaries instead of thirty-two beside the 60 registers of the prefetched
    // halo: the lane's sums and squares over the tile's 4 rows (the values as stored: rounded to bf16 once more here), then
    // a reduce-scatter over the 32 lanes of the half (col = voxel): xor 16 keeps sums | squares, two mirror steps halve
    // the four channels down to one, two butterfly steps inside the quad finish it: lane (b4, b3, b2) holds {sum | sumsq by b4} of channel
    // 8 gq + 4 half + 2 b3 + b2 and adds it to its running value of that quad.
    const bool vox_ok = od < g.D && ow < g.W;
    const bool tile_full = od < g.D && h0 + MT <= g.H && w0 + kLtW <= g.W;          // wave-uniform
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      if (8 * gq >= g.Cout) continue;                 // (wave-uniform; the quad of the other half may still be live)
      const int co = 8 * gq + 4 * half;
      float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
      if (tile_full && 8 * gq + 8 <= g.Cout) {          // the common case, straight-line: every voxel and both halves' channels live
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[t][4 * gq + e] + (bias ? bias[co + e] : 0.f);
          const unsigned p0 = pack2_bf16(o[0], o[1]), p1 = pack2_bf16(o[2], o[3]);
          const float r[4] = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u), __uint_as_float(p1 << 16),
                              __uint_as_float(p1 & 0xffff0000u)};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s4[e] += r[e];
            q4[e] = __builtin_fmaf(r[e], r[e], q4[e]);
          }
        }
      } else if (vox_ok && co < g.Cout) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          if (h0 + t >= g.H) continue;
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[t][4 * gq + e] + (bias ? bias[co + e] : 0.f);
          const unsigned p0 = pack2_bf16(o[0], o[1]), p1 = pack2_bf16(o[2], o[3]);
          const float r[4] = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u), __uint_as_float(p1 << 16),
                              __uint_as_float(p1 & 0xffff0000u)};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s4[e] += r[e];
            q4[e] += r[e] * r[e];
          }
        }
      }
      float v4[4];
      {
        const bool up = (col & 16) != 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = (up ? q4[e] : s4[e]) + __shfl_xor(up ? s4[e] : q4[e], 16, 64);
      }
      // inside a row of 16 lanes the partners are the DPP mirrors (lane i <-> 15 - i has the other bit 3, i <-> 7 - i inside
      // eight lanes the other bit 2): VALU moves instead of ds_bpermute round trips (36 of them per tile cost the 24 -> 24
      // layer 0.17 ms)
#define TRANSOAR_DPP_F(v, CTRL) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true))
      float v2[2];
      {
        const bool up = (col & 8) != 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float give = up ? v4[e] : v4[2 + e];
          v2[e] = (up ? v4[2 + e] : v4[e]) + TRANSOAR_DPP_F(give, 0x140);          // row_mirror
        }
      }
      float v1;
      {
        const bool up = (col & 4) != 0;
        const float give = up ? v2[0] : v2[1];
        v1 = (up ? v2[1] : v2[0]) + TRANSOAR_DPP_F(give, 0x141);                    // row_half_mirror
      }
      v1 += TRANSOAR_DPP_F(v1, 0x4E);                                               // quad_perm [2,3,0,1]
      v1 += TRANSOAR_DPP_F(v1, 0xB1);                                               // quad_perm [1,0,3,2]
#undef TRANSOAR_DPP_F
      srun[gq] += v1;
    }
  }
  __syncthreads();          // every wave is done with this halo before the next one lands
  }
  if (STATS) {
    // lane (half, b4, b3, b2): srun[gq] = {sum | sumsq by b4} of output channel 8 gq + 4 half + 2 b3 + b2; the four waves
    // (depths) meet in LDS
    float* red = reinterpret_cast<float*>(halo);              // [wave][which][32 channels]
    if ((col & 3) == 0) {
      const int which = (col >> 4) & 1;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        red[(wave * 2 + which) * 32 + 8 * gq + 4 * half + 2 * ((col >> 3) & 1) + ((col >> 2) & 1)] = srun[gq];
    }
    __syncthreads();
    if (threadIdx.x < 64)
      stat_part[trow * 64 + threadIdx.x] = red[threadIdx.x] + red[64 + threadIdx.x] + red[128 + threadIdx.x] + red[192 + threadIdx.x];
  }
}

// ---------------------------------------------------------------------------
// weight gradient, K = voxels.
//   gyT  (Cout, V)       bf16, V = N*Do*Ho*Wo, voxel-contiguous (channels first)
//   xT3  (3, Cin, N, D, H, W) bf16: xT3[kw][ci][n][d][h][w] = x[n][d][h][w+kw-1][ci]
//        (zero outside) -- the three W-shifts are materialised so every fragment
//        is an aligned 16-byte load.  stride 1 only in this layout (stride-2
//        layers pass a W-decimated copy, see the host shim).
//   dw   (27, Cout, CinP) fp32, accumulated with atomics over the voxel split.
// D[i = cout][j = (tap,ci)] tiles; a wave owns NT column tiles and walks a
// range of voxel rows (n, d, h) in steps of 16 voxels along W.
// ---------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void conv3d_k3_wgrad(const unsigned short* __restrict__ gyT,
                                                       const unsigned short* __restrict__ xT3,
                                                       float* __restrict__ dw, ConvGeom g,
                                                       int rows_per_block, unsigned gy_bytes,
                                                       unsigned x_bytes) {
  // geometry here: output grid (Do,Ho,Wo) == the gy grid; input grid (D,H,Wd) where Wd is
  // the (possibly decimated) W extent of xT3 rows that line up 1:1 with gy columns.
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col = lane & 31;
  const int half = lane >> 5;
  const int n_cols = 27 * g.Cin;                       // (tap, ci) columns
  const int col_tiles = (n_cols + 31) >> 5;
  const int ct0 = (blockIdx.y * 4 + wave) * NT;        // first column tile of this wave
  if (ct0 >= col_tiles) return;
  const int co_base = blockIdx.z * 32;

  const __amdgpu_buffer_rsrc_t gr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(gyT), 0, static_cast<int>(gy_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t xr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xT3), 0, static_cast<int>(x_bytes), 0x00020000);

  const long V = static_cast<long>(g.N) * g.Do * g.Ho * g.Wo;
  const long plane = static_cast<long>(g.N) * g.D * g.H * g.W;   // voxels per channel of xT3

  // column -> (tap, ci): fixed per lane per tile
  int kd[NT], kh[NT], kwv[NT], ci[NT];
  bool col_ok[NT];
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    const int c = (ct0 + a) * 32 + col;
    col_ok[a] = c < n_cols;
    const int cc = col_ok[a] ? c : 0;
    const int tap = cc / g.Cin;
    ci[a] = cc - tap * g.Cin;
    kd[a] = tap / 9;
    kh[a] = (tap / 3) % 3;
    kwv[a] = tap % 3;
  }
  const int co = co_base + col;
  const bool co_ok = co < g.Cout;

  f32x16 acc[NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  const int n_rows = g.N * g.Do * g.Ho;                // (n, od, oh) rows of Wo voxels
  const int row0 = blockIdx.x * rows_per_block;
  const int row1 = min(row0 + rows_per_block, n_rows);
  for (int row = row0; row < row1; ++row) {
    const int oh = row % g.Ho;
    const int r2 = row / g.Ho;
    const int od = r2 % g.Do;
    const int n = r2 / g.Do;
    // input row for each column tile's tap
    long xrow[NT];
    bool row_ok[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const int id = od * g.stride + kd[a] - 1, ih = oh * g.stride + kh[a] - 1;
      row_ok[a] = col_ok[a] && static_cast<unsigned>(id) < static_cast<unsigned>(g.D) &&
                  static_cast<unsigned>(ih) < static_cast<unsigned>(g.H);
      xrow[a] = (static_cast<long>(kwv[a]) * g.Cin + ci[a]) * plane +
                ((static_cast<long>(n) * g.D + id) * g.H + ih) * g.W;
    }
    const long grow = static_cast<long>(co) * V + static_cast<long>(row) * g.Wo;
    for (int w0 = 0; w0 < g.Wo; w0 += 16) {
      const int wk = w0 + 8 * half;                    // this lane-half's 8 voxels
      const bool w_ok = wk < g.Wo;                     // Wo % 8 == 0 (host check)
      const unsigned goff = (co_ok && w_ok) ? static_cast<unsigned>((grow + wk) * 2) : kOOB;
      const bf16x8 gf = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(gr, goff, 0, 0));
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        const unsigned xoff = (row_ok[a] && w_ok) ? static_cast<unsigned>((xrow[a] + wk) * 2) : kOOB;
        const bf16x8 xf = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xr, xoff, 0, 0));
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf, xf, acc[a], 0, 0, 0);
      }
    }
  }

  // D[i = cout][j = column]; lane: column j = col, rows (r&3) + 8*(r>>2) + 4*half
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    if (!col_ok[a]) continue;
    const int tap = kd[a] * 9 + kh[a] * 3 + kwv[a];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = co_base + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (o < g.Cout)
        __hip_atomic_fetch_add(dw + (static_cast<long>(tap) * g.Cout + o) * g.CinP + ci[a], acc[a][r],
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------------------
// Cin == 1 stencil: x (N,D,H,W) bf16, w (27, Cout) fp32, y (N,D,H,W,Cout) bf16.
// One thread per output voxel: its 27 inputs sit in registers, the weights are
// wave-uniform (scalar loads), output channels are produced 8 at a time.
// ---------------------------------------------------------------------------
constexpr int kC1MaxCout = 64;
__global__ __launch_bounds__(256) void conv3d_c1_fwd(const unsigned short* __restrict__ x,
                                                     const float* __restrict__ w,
                                                     unsigned short* __restrict__ y, int N, int D, int H,
                                                     int W, int Cout, long n_vox) {
  // the 256 voxels of a workgroup are consecutive rows of y: their Cout-channel rows are staged in
  // LDS in the global order and leave as whole 16-byte-per-lane contiguous stores (a thread writing
  // its own row scatters 16-byte pieces at a Cout*2-byte stride: 400 GB/s measured on the 630 MB output)
  __shared__ u32x4c stage[256 * kC1MaxCout / 8];
  const long v0 = static_cast<long>(blockIdx.x) * 256;
  const long v = v0 + threadIdx.x;
  const bool live = v < n_vox;
  const long vc = live ? v : n_vox - 1;
  const int ow = static_cast<int>(vc % W);
  const long r1 = vc / W;
  const int oh = static_cast<int>(r1 % H);
  const long r2 = r1 / H;
  const int od = static_cast<int>(r2 % D);
  const long nbase = (r2 / D) * D;
  float xv[27];
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int id = od + kd - 1, ih = oh + kh - 1, iw = ow + kw - 1;
        const bool ok = static_cast<unsigned>(id) < static_cast<unsigned>(D) &&
                        static_cast<unsigned>(ih) < static_cast<unsigned>(H) &&
                        static_cast<unsigned>(iw) < static_cast<unsigned>(W);
        // branch-free: a clamped (always valid) address, then a select -- under a divergent branch
        // each of the 27 loads waits for its own round trip
        const int cd = min(max(id, 0), D - 1), chh = min(max(ih, 0), H - 1), cw = min(max(iw, 0), W - 1);
        const float val = bf2f(x[((nbase + cd) * H + chh) * W + cw]);
        xv[kd * 9 + kh * 3 + kw] = ok ? val : 0.f;
      }
  const int cpv = Cout >> 3;                  // 16-byte chunks per voxel row
  for (int c0 = 0; c0 < Cout; c0 += 8) {      // uniform loop: weights come through the scalar cache
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const float* wt = w + t * Cout + c0;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += xv[t] * wt[e];
    }
    u32x4c pk;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      pk[e] = pack2_bf16(acc[2 * e], acc[2 * e + 1]);
    stage[threadIdx.x * cpv + (c0 >> 3)] = pk;
  }
  __syncthreads();
  const long chunks = min(static_cast<long>(256), n_vox - v0) * cpv;
  u32x4c* dst = reinterpret_cast<u32x4c*>(y + v0 * Cout);
  for (int i = threadIdx.x; i < chunks; i += 256) dst[i] = stage[i];
}

// ---------------------------------------------------------------------------
// Cin == 1 weight gradient:  dW[cout][tap] = sum_voxels dy[voxel][cout] * x[voxel + tap]
//   x (N,D,H,W) bf16 ; dy (N,D,H,W,Cout) bf16 (NDHWC) ; Cout <= 32 ; W % 16 == 0
//   partial (gridDim.x, 32, 32) fp32: per workgroup the 32x32 (cout x tap) tile, rows >= Cout and
//   columns >= 27 zero; the caller sums over workgroups.
// One v_mfma_f32_32x32x16_bf16 per 16 consecutive voxels of a W-row: A = dy^T (cout x voxel),
// B = the 27 shifted views of x (voxel x tap).  Both operands are gathered with 2-byte loads (8 per
// lane each; the voxel axis is the slow one in memory) -- the kernel is bound by that, not by the
// MFMA: 630 MB of dy once, x from L2.  (aten's fallback for this layer is im2col + a 16x16-tile
// GEMM per sample: 3.9 ms; this: see profiles/.)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv3d_c1_wgrad(const unsigned short* __restrict__ x,
                                                       const unsigned short* __restrict__ dy,
                                                       float* __restrict__ partial, int N, int D, int H, int W,
                                                       int Cout, long n_rows, int rows_per_wave) {
  __shared__ float red[4][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, kg = lane >> 5;
  // B side: this lane's tap
  const int tap = col;
  const bool tap_ok = tap < 27;
  const int kd = tap / 9 - 1, kh = (tap / 3) % 3 - 1, kw = tap % 3 - 1;
  // A side: this lane's output channel
  const bool co_ok = col < Cout;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const long row0 = (static_cast<long>(blockIdx.x) * 4 + wave) * rows_per_wave;
  for (int rr = 0; rr < rows_per_wave; ++rr) {
    const long row = row0 + rr;                        // (b*D + d)*H + h
    if (row >= n_rows) break;
    const int h = static_cast<int>(row % H);
    const long bd = row / H;
    const int d = static_cast<int>(bd % D);
    const int id = d + kd, ih = h + kh;
    const bool row_ok = tap_ok && static_cast<unsigned>(id) < static_cast<unsigned>(D) &&
                        static_cast<unsigned>(ih) < static_cast<unsigned>(H);
    const unsigned short* xrow = x + ((bd - d + (row_ok ? id : d)) * H + (row_ok ? ih : h)) * W;   // clamped: always valid
    const unsigned short* dyrow = dy + row * W * Cout + (co_ok ? col : 0);
    for (int w0 = 0; w0 < W; w0 += 16) {
      unsigned short av[8], bv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int w = w0 + 8 * kg + j;
        av[j] = dyrow[static_cast<long>(w) * Cout];
        const int iw = w + kw;
        const unsigned short xv = xrow[min(max(iw, 0), W - 1)];
        bv[j] = (row_ok && static_cast<unsigned>(iw) < static_cast<unsigned>(W)) ? xv : static_cast<unsigned short>(0);
      }
      u32x4c ap, bp;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ap[j] = co_ok ? (static_cast<unsigned>(av[2 * j]) | (static_cast<unsigned>(av[2 * j + 1]) << 16)) : 0u;
        bp[j] = static_cast<unsigned>(bv[2 * j]) | (static_cast<unsigned>(bv[2 * j + 1]) << 16);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ap), __builtin_bit_cast(bf16x8, bp), acc,
                                                    0, 0, 0);
    }
  }
  // 4 waves -> one tile: D[row = (r&3) + 8*(r>>2) + 4*kg][col]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  if (wave == 0) {
    float* out = partial + static_cast<long>(blockIdx.x) * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane];
      out[((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + col] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// The same weight gradient (Cin == 1) with coalesced loads: conv3d_c1_wgrad above gathers both MFMA operands with
// 2-byte loads (16 per lane and MFMA) and waits for them -- 0.76 ms on the flagship stem where the 630 MB of dy are
// 0.15 ms of HBM time (profiles/r03_conv_pmc.txt: 181 K of 204 K wave cycles waiting).  Here a workgroup walks W-rows
// (b, d, h) of W <= 256 voxels:
//   dy row   -> LDS as it lies in memory ([voxel][Cout], 64-byte pitch, 16-byte loads one row ahead); the A fragment
//               (cout x 16 voxels) is cut out with the transposing ds_read_b64_tr_b16 (4 rows x 64 bytes = all 64 banks)
//   x rows   -> the 9 (kd, kh) neighbour rows, each stored THREE times, shifted by kw - 1 voxels (built in registers
//               from the aligned 16-byte load + the neighbour lanes' edge elements): the B fragment of lane (tap, kg)
//               is ONE aligned ds_read_b128 of copy kw; taps 27..31 read a row of zeros
//   one v_mfma_f32_32x32x16_bf16 per 16 voxels and wave; the four waves take 64 voxels of the row each.
// partial (gridDim.x, 32, 32) as above.
// ---------------------------------------------------------------------------
typedef short c1_s16x4 __attribute__((ext_vector_type(4)));
typedef short c1_s16x8 __attribute__((ext_vector_type(8)));
constexpr int kC1XP = 256 + 16;              // elements per staged x row

template <int CO8>
__global__ __launch_bounds__(256) void conv3d_c1_wgrad_tr(const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy,
                                                          float* __restrict__ partial, int N, int D, int H, int W, int Cout,
                                                          long n_rows, int rows_per_wg, unsigned x_bytes, unsigned dy_bytes) {
  __shared__ __attribute__((aligned(16))) unsigned char dyt[256 * 64 + 512];       // + the transposing reads' over-reach; reused for the final sum
  __shared__ __attribute__((aligned(16))) unsigned short xs[3][10][kC1XP];         // [kw][kd * 3 + kh | 9 = zeros][w]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kg = lane >> 5;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(x), 0, static_cast<int>(x_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(dy), 0, static_cast<int>(dy_bytes), 0x00020000);
  for (int i = tid; i < (256 * 64 + 512) / 4; i += 256) reinterpret_cast<unsigned*>(dyt)[i] = 0u;       // channels >= Cout stay zero
  for (int i = tid; i < 3 * 10 * kC1XP / 2; i += 256) reinterpret_cast<unsigned*>(&xs[0][0][0])[i] = 0u;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // fragment addresses (constant over the rows)
  const int t_row = 8 * kg + ((lane & 15) >> 2), t_ch = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const unsigned char* a_base = dyt + (64 * wave + t_row) * 64 + t_ch * 2;
  const int tap = lane & 31;
  const unsigned short* b_base = tap < 27 ? &xs[tap % 3][tap / 3][64 * wave + 8 * kg] : &xs[0][9][64 * wave + 8 * kg];
  typedef __attribute__((address_space(3))) c1_s16x4 lds_s16x4;

  const long row_beg = static_cast<long>(blockIdx.x) * rows_per_wg, row_end = min(n_rows, row_beg + rows_per_wg);
  u32x4c pdy[CO8], px[2];
  const int dy_pieces = W * CO8;                       // 16-byte pieces of a dy row, contiguous in memory
  auto prefetch = [&](long row) {
#pragma unroll
    for (int k = 0; k < CO8; ++k) {
      const int p = tid + 256 * k;
      pdy[k] = __builtin_amdgcn_raw_buffer_load_b128(rdy, p < dy_pieces ? static_cast<unsigned>(row) * static_cast<unsigned>(dy_pieces * 16) + p * 16 : 0x80000000u, 0, 0);
    }
    const int h = static_cast<int>(row % H);
    const long bd = row / H;
    const int d = static_cast<int>(bd % D);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + 256 * k, r = i >> 5, piece = i & 31;
      const int id = d + r / 3 - 1, ih = h + r % 3 - 1;
      const bool ok = r < 9 && piece * 8 < W && static_cast<unsigned>(id) < static_cast<unsigned>(D) && static_cast<unsigned>(ih) < static_cast<unsigned>(H);
      px[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? static_cast<unsigned>(((bd - d + id) * H + ih) * W + piece * 8) * 2u : 0x80000000u, 0, 0);
    }
  };
  if (row_beg < row_end) prefetch(row_beg);
  __syncthreads();
  for (long row = row_beg; row < row_end; ++row) {
    // ---- registers -> LDS
#pragma unroll
    for (int k = 0; k < CO8; ++k) {
      const int p = tid + 256 * k;
      if (p < dy_pieces) *reinterpret_cast<u32x4c*>(dyt + (p / CO8) * 64 + (p % CO8) * 16) = pdy[k];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + 256 * k, r = i >> 5, piece = i & 31;
      const u32x4c v = px[k];
      // edge elements of the neighbouring pieces of the same row (32 consecutive lanes); zeros beyond the row
      unsigned left = __shfl_up(v[3], 1), right = __shfl_down(v[0], 1);
      if (piece == 0) left = 0u;
      if (piece * 8 + 8 >= W) right = 0u;
      if (r < 9 && piece * 8 < W) {
        u32x4c lo, hi;                                  // shifted by -1 / +1 voxel
        lo[0] = __builtin_amdgcn_alignbit(v[0], left, 16); lo[1] = __builtin_amdgcn_alignbit(v[1], v[0], 16);
        lo[2] = __builtin_amdgcn_alignbit(v[2], v[1], 16); lo[3] = __builtin_amdgcn_alignbit(v[3], v[2], 16);
        hi[0] = __builtin_amdgcn_alignbit(v[1], v[0], 16); hi[1] = __builtin_amdgcn_alignbit(v[2], v[1], 16);
        hi[2] = __builtin_amdgcn_alignbit(v[3], v[2], 16); hi[3] = __builtin_amdgcn_alignbit(right, v[3], 16);
        *reinterpret_cast<u32x4c*>(&xs[0][r][piece * 8]) = lo;       // tap kw = 0 sees x[w - 1]
        *reinterpret_cast<u32x4c*>(&xs[1][r][piece * 8]) = v;
        *reinterpret_cast<u32x4c*>(&xs[2][r][piece * 8]) = hi;
      }
    }
    __syncthreads();
    if (row + 1 < row_end) prefetch(row + 1);           // in flight during the MFMAs
    if (64 * wave < W) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const c1_s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a_base + ks * 16 * 64));
        const c1_s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a_base + ks * 16 * 64 + 4 * 64));
        const c1_s16x8 a = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
        const c1_s16x8 b = *reinterpret_cast<const c1_s16x8*>(b_base + ks * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // 4 waves -> one tile: D[row = cout (r & 3) + 8 (r >> 2) + 4 kg][col = tap]
  float* red = reinterpret_cast<float*>(dyt);            // [wave][16][64]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (wave == 0) {
    float* out = partial + static_cast<long>(blockIdx.x) * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = (r & 3) + 8 * (r >> 2) + 4 * kg;
      const float v = red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane] + red[(48 + r) * 64 + lane];
      out[co * 32 + tap] = (co < Cout && tap < 27) ? v : 0.f;
    }
  }
}

#include "conv3d_wgrad_lds.hpp"
#include "conv3d_wgrad_tr.hpp"

// ---------------------------------------------------------------------------
// layout changes between channels-last (N, V, C) and channels-first (N, C, V),
// bf16.  One thread per voxel: 16-byte accesses on the channels-last side
// (8 channels of its voxel), 2-byte accesses coalesced across the wave on the
// channels-first side.  (torch's generic strided copy does the same job at
// ~100 GB/s; these run near HBM speed.)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layout_vc_to_cv(const unsigned short* __restrict__ in,
                                                        unsigned short* __restrict__ out, long V, int C) {
  const long v = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (v >= V) return;
  const long n = blockIdx.y;
  const unsigned short* src = in + (n * V + v) * C;
  unsigned short* dst = out + n * C * V + v;
  for (int c = 0; c < C; c += 8) {
    const u32x4c r = *reinterpret_cast<const u32x4c*>(src + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dst[static_cast<long>(c + 2 * e) * V] = static_cast<unsigned short>(r[e] & 0xffffu);
      dst[static_cast<long>(c + 2 * e + 1) * V] = static_cast<unsigned short>(r[e] >> 16);
    }
  }
}

__global__ __launch_bounds__(256) void layout_cv_to_vc(const unsigned short* __restrict__ in,
                                                        unsigned short* __restrict__ out, long V, int C) {
  const long v = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (v >= V) return;
  const long n = blockIdx.y;
  const unsigned short* src = in + n * C * V + v;
  unsigned short* dst = out + (n * V + v) * C;
  for (int c = 0; c < C; c += 8) {
    u32x4c r;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      r[e] = static_cast<unsigned>(src[static_cast<long>(c + 2 * e) * V]) |
             (static_cast<unsigned>(src[static_cast<long>(c + 2 * e + 1) * V]) << 16);
    *reinterpret_cast<u32x4c*>(dst + c) = r;
  }
}

static inline bool fits32(long bytes) { return bytes > 0 && bytes < 0xfffffff0L; }

}  // namespace transoar

using namespace transoar;

extern "C" int transoar_conv3d_k3_forward(const void* x, const void* wk, const float* bias, void* y, int N,
                                          int D, int H, int W, int Cin, int Cout, int stride, int dilated_input,
                                          void* hip_stream) {
  if (!x || !wk || !y) return TRANSOAR_CONV_ERR_NULL;
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return TRANSOAR_CONV_ERR_DIM;
  if ((Cin & 7) || (Cout & 3)) return TRANSOAR_CONV_ERR_CHANNELS;
  if (!(stride == 1 || stride == 2) || (dilated_input && stride != 1)) return TRANSOAR_CONV_ERR_DIM;
  ConvGeom g;
  g.N = N; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.CinP = (Cin + 7) & ~7; g.stride = stride;
  if (dilated_input) {       // output grid = 2x the input grid (transposed conv of a stride-2 layer)
    g.Do = 2 * D; g.Ho = 2 * H; g.Wo = 2 * W;
  } else {
    g.Do = (D - 1) / stride + 1; g.Ho = (H - 1) / stride + 1; g.Wo = (W - 1) / stride + 1;
  }
  const long n_vox = static_cast<long>(N) * g.Do * g.Ho * g.Wo;
  const long x_bytes = static_cast<long>(N) * D * H * W * Cin * 2;
  const long w_bytes = 27L * Cout * g.CinP * 2;
  if (!fits32(x_bytes) || !fits32(w_bytes) || n_vox >= (1L << 31)) return TRANSOAR_CONV_ERR_DIM;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  auto xp = static_cast<const unsigned short*>(x);
  auto wp = static_cast<const unsigned short*>(wk);
  auto yp = static_cast<unsigned short*>(y);
  // full-resolution stride-1 layers with few channels: halo tile in LDS (conv3d_k3_lds)
  static const bool no_lds = getenv("TRANSOAR_CONV_NO_LDS") != nullptr;
  if (!no_lds && !dilated_input && stride == 1 && Cout <= 32 && (Cin == 8 || Cin == 16 || Cin == 24) &&
      static_cast<long>(D) * H * W >= (1L << 16)) {
    const int tiles_d = (D + kLtD - 1) / kLtD, tiles_h = (H + kLtH - 1) / kLtH, tiles_w = (W + kLtW - 1) / kLtW;
    const long n_tiles = static_cast<long>(N) * tiles_d * tiles_h;      // rows of tiles_w tiles: one workgroup each
    const dim3 lgrid(static_cast<unsigned>(((n_tiles + 7) / 8) * 8));
#define TRANSOAR_CONV_LDS(CPTV)                                                                              \
  hipLaunchKernelGGL((conv3d_k3_lds<CPTV>), lgrid, dim3(256), 0, st, xp, wp, bias, yp, g, tiles_d, tiles_h, tiles_w, \
                     n_tiles, static_cast<unsigned>(x_bytes), static_cast<unsigned>(w_bytes))
    if (Cin == 8) TRANSOAR_CONV_LDS(1); else if (Cin == 16) TRANSOAR_CONV_LDS(2); else TRANSOAR_CONV_LDS(3);
#undef TRANSOAR_CONV_LDS
    return static_cast<int>(hipGetLastError());
  }
  // register blocking per wave: 4 voxel tiles x 1 cout tile (Cout <= 32) or 2 x 2 (64 accumulator
  // registers either way, 3 waves per SIMD to cover the load latency)
  const bool wide = Cout > 32;
  const int MT = wide ? 2 : 4, NT = wide ? 2 : 1;
  const dim3 grid(static_cast<unsigned>((n_vox + 4 * 32 * MT - 1) / (4 * 32 * MT)),
                  static_cast<unsigned>((Cout + 32 * NT - 1) / (32 * NT)));
#define TRANSOAR_CONV(MTV, NTV, DILV)                                                                    \
  hipLaunchKernelGGL((conv3d_k3_igemm<MTV, NTV, DILV>), grid, dim3(256), 0, st, xp, wp, bias, yp, g, n_vox, \
                     static_cast<unsigned>(x_bytes), static_cast<unsigned>(w_bytes))
  if (dilated_input) { if (wide) TRANSOAR_CONV(2, 2, true); else TRANSOAR_CONV(4, 1, true); }
  else { if (wide) TRANSOAR_CONV(2, 2, false); else TRANSOAR_CONV(4, 1, false); }
#undef TRANSOAR_CONV
  return static_cast<int>(hipGetLastError());
}

// Cin = 1 stem through the LDS kernel without materialising a channel-padded copy of the volume:
// x (N, D, H, W) bf16, wk (27, Cout, 8) with the one real input channel first (the other 7 are multiplied by the
// zeros the staging writes).  Stride 1, Cout <= 32.
extern "C" int transoar_conv3d_k3_forward_c1(const void* x, const void* wk, const float* bias, void* y, int N, int D,
                                             int H, int W, int Cout, void* hip_stream) {
  if (!x || !wk || !y) return TRANSOAR_CONV_ERR_NULL;
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0) return TRANSOAR_CONV_ERR_DIM;
  if ((Cout & 3) || Cout > 32) return TRANSOAR_CONV_ERR_CHANNELS;
  ConvGeom g;
  g.N = N; g.D = D; g.H = H; g.W = W; g.Cin = 8; g.Cout = Cout; g.CinP = 8; g.stride = 1;
  g.Do = D; g.Ho = H; g.Wo = W;
  const long x_bytes = static_cast<long>(N) * D * H * W * 2;
  const long w_bytes = 27L * Cout * 8 * 2;
  if (!fits32(x_bytes) || static_cast<long>(N) * D * H * W >= (1L << 31)) return TRANSOAR_CONV_ERR_DIM;
  const int tiles_d = (D + kLtD - 1) / kLtD, tiles_h = (H + kLtH - 1) / kLtH, tiles_w = (W + kLtW - 1) / kLtW;
  const long n_rows = static_cast<long>(N) * tiles_d * tiles_h;
  hipLaunchKernelGGL((conv3d_k3_lds<1, true>), dim3(static_cast<unsigned>(((n_rows + 7) / 8) * 8)), dim3(256), 0,
                     static_cast<hipStream_t>(hip_stream), static_cast<const unsigned short*>(x),
                     static_cast<const unsigned short*>(wk), bias, static_cast<unsigned short*>(y), g, tiles_d, tiles_h,
                     tiles_w, n_rows, static_cast<unsigned>(x_bytes), static_cast<unsigned>(w_bytes));
  return static_cast<int>(hipGetLastError());
}

// ---- forward with the InstanceNorm statistics in the epilogue (conv3d_k3_lds<.., STATS = true>)
extern "C" int transoar_conv3d_k3_stat_rows(int N, int D, int H, int W, int Cin, int Cout, int stride) {
  static const bool no_lds = getenv("TRANSOAR_CONV_NO_LDS") != nullptr;
  if (no_lds || stride != 1 || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout > 32 || (Cout & 3)) return 0;
  if (!(Cin == 1 || Cin == 8 || Cin == 16 || Cin == 24) || static_cast<long>(D) * H * W < (1L << 16)) return 0;
  const long rows = static_cast<long>(N) * ((D + kLtD - 1) / kLtD) * ((H + kLtH - 1) / kLtH);
  return rows < (1L << 30) ? static_cast<int>(rows) : 0;
}

extern "C" int transoar_conv3d_k3_forward_stats(const void* x, const void* wk, const float* bias, void* y, float* stat_part,
                                                int N, int D, int H, int W, int Cin, int Cout, void* hip_stream) {
  if (!x || !wk || !y || !stat_part) return TRANSOAR_CONV_ERR_NULL;
  if (transoar_conv3d_k3_stat_rows(N, D, H, W, Cin, Cout, 1) == 0) return TRANSOAR_CONV_ERR_DIM;
  ConvGeom g;
  g.N = N; g.D = D; g.H = H; g.W = W; g.Cin = Cin == 1 ? 8 : Cin; g.Cout = Cout; g.CinP = g.Cin; g.stride = 1;
  g.Do = D; g.Ho = H; g.Wo = W;
  const long x_bytes = static_cast<long>(N) * D * H * W * Cin * 2;
  const long w_bytes = 27L * Cout * g.CinP * 2;
  if (!fits32(x_bytes) || !fits32(w_bytes) || static_cast<long>(N) * D * H * W >= (1L << 31)) return TRANSOAR_CONV_ERR_DIM;
  const int tiles_d = (D + kLtD - 1) / kLtD, tiles_h = (H + kLtH - 1) / kLtH, tiles_w = (W + kLtW - 1) / kLtW;
  const long n_rows = static_cast<long>(N) * tiles_d * tiles_h;
  const dim3 grid(static_cast<unsigned>(((n_rows + 7) / 8) * 8));
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  auto xp = static_cast<const unsigned short*>(x);
  auto wp = static_cast<const unsigned short*>(wk);
  auto yp = static_cast<unsigned short*>(y);
#define TRANSOAR_CONV_LDS_STATS(CPTV, C1V)                                                                             \
  hipLaunchKernelGGL((conv3d_k3_lds<CPTV, C1V, true>), grid, dim3(256), 0, st, xp, wp, bias, yp, g, tiles_d, tiles_h, tiles_w, \
                     n_rows, static_cast<unsigned>(x_bytes), static_cast<unsigned>(w_bytes), stat_part)
  if (Cin == 1) TRANSOAR_CONV_LDS_STATS(1, true);
  else if (Cin == 8) TRANSOAR_CONV_LDS_STATS(1, false);
  else if (Cin == 16) TRANSOAR_CONV_LDS_STATS(2, false);
  else TRANSOAR_CONV_LDS_STATS(3, false);
#undef TRANSOAR_CONV_LDS_STATS
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_conv3d_k3_wgrad(const void* gyT, const void* xT3, float* dw, int N, int D, int H, int W,
                                        int Cin, int Do, int Ho, int Wo, int Cout, int stride_dh,
                                        void* hip_stream) {
  if (!gyT || !xT3 || !dw) return TRANSOAR_CONV_ERR_NULL;
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Wo & 7) || W != Wo)
    return TRANSOAR_CONV_ERR_DIM;
  ConvGeom g;
  g.N = N; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.CinP = (Cin + 7) & ~7;
  g.Do = Do; g.Ho = Ho; g.Wo = Wo; g.stride = stride_dh;
  const long gy_bytes = static_cast<long>(Cout) * N * Do * Ho * Wo * 2;
  const long x_bytes = 3L * Cin * N * D * H * W * 2;
  if (!fits32(gy_bytes) || !fits32(x_bytes)) return TRANSOAR_CONV_ERR_DIM;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  constexpr int NT = 4;
  const int col_tiles = (27 * Cin + 31) / 32;
  const int n_rows = N * Do * Ho;
  // split the voxel rows so the grid has a few thousand workgroups
  const int wave_groups = (col_tiles + 4 * NT - 1) / (4 * NT);
  const int co_tiles = (Cout + 31) / 32;
  int splits = 4096 / (wave_groups * co_tiles);
  splits = splits < 1 ? 1 : (splits > n_rows ? n_rows : splits);
  const int rows_per_block = (n_rows + splits - 1) / splits;
  const dim3 grid(static_cast<unsigned>((n_rows + rows_per_block - 1) / rows_per_block),
                  static_cast<unsigned>(wave_groups), static_cast<unsigned>(co_tiles));
  hipLaunchKernelGGL((conv3d_k3_wgrad<NT>), grid, dim3(256), 0, st, static_cast<const unsigned short*>(gyT),
                     static_cast<const unsigned short*>(xT3), dw, g, rows_per_block,
                     static_cast<unsigned>(gy_bytes), static_cast<unsigned>(x_bytes));
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_conv3d_k3_wgrad_lds(const void* x, const void* dy, float* partial, int n_wg, int N, int D, int H,
                                            int W, int Cin, int Cout, int ci0, int ci_n, int co0, int co_n,
                                            void* hip_stream) {
  if (!x || !dy || !partial) return TRANSOAR_CONV_ERR_NULL;
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || n_wg <= 0) return TRANSOAR_CONV_ERR_DIM;
  if (Cin <= 0 || Cout <= 0 || (Cin & 7) || (Cout & 7) || (W & 63)) return TRANSOAR_CONV_ERR_CHANNELS;
  if (ci_n <= 0 || co_n <= 0 || ci_n > 32 || co_n > 32 || (ci_n & 7) || (co_n & 7) || (ci0 & 7) || (co0 & 7) ||
      ci0 < 0 || co0 < 0 || ci0 + ci_n > Cin || co0 + co_n > Cout)
    return TRANSOAR_CONV_ERR_CHANNELS;
  const long n_units = static_cast<long>(N) * D * H * (W / 64);
  const int per = static_cast<int>((n_units + n_wg - 1) / n_wg);
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  auto xs = static_cast<const unsigned short*>(x);
  auto ds = static_cast<const unsigned short*>(dy);
#define TRANSOAR_WG_CASE(CI, CO)                                                                         \
  if (ci_n == 8 * CI && co_n == 8 * CO) {                                                                \
    hipLaunchKernelGGL((conv3d_k3_wgrad_lds<CI, CO>), dim3(static_cast<unsigned>(n_wg)), dim3(kWgThreads), 0, st, xs, ds, \
                       partial, N, D, H, W, n_units, per, Cin, ci0, Cout, co0);                          \
    return static_cast<int>(hipGetLastError());                                                          \
  }
  TRANSOAR_WG_CASE(1, 1) TRANSOAR_WG_CASE(1, 2) TRANSOAR_WG_CASE(1, 3) TRANSOAR_WG_CASE(1, 4)
  TRANSOAR_WG_CASE(2, 1) TRANSOAR_WG_CASE(2, 2) TRANSOAR_WG_CASE(2, 3) TRANSOAR_WG_CASE(2, 4)
  TRANSOAR_WG_CASE(3, 1) TRANSOAR_WG_CASE(3, 2) TRANSOAR_WG_CASE(3, 3) TRANSOAR_WG_CASE(3, 4)
  TRANSOAR_WG_CASE(4, 1) TRANSOAR_WG_CASE(4, 2) TRANSOAR_WG_CASE(4, 3) TRANSOAR_WG_CASE(4, 4)
#undef TRANSOAR_WG_CASE
  return TRANSOAR_CONV_ERR_CHANNELS;
}

extern "C" int transoar_conv3d_k3_wgrad_tr(const void* x, const void* dy, float* partial, int n_wg, int N, int D, int H,
                                           int W, int Cin, int Cout, int ci0, int ci_n, int co0, int co_n,
                                           void* hip_stream) {
  if (!x || !dy || !partial) return TRANSOAR_CONV_ERR_NULL;
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || n_wg <= 0) return TRANSOAR_CONV_ERR_DIM;
  if (Cin <= 0 || Cout <= 0 || (Cin & 7) || (Cout & 7) || (W & 63)) return TRANSOAR_CONV_ERR_CHANNELS;
  if (ci_n <= 0 || co_n <= 0 || ci_n > 32 || co_n > 32 || (ci_n & 7) || (co_n & 7) || (ci0 & 7) || (co0 & 7) ||
      ci0 < 0 || co0 < 0 || ci0 + ci_n > Cin || co0 + co_n > Cout)
    return TRANSOAR_CONV_ERR_CHANNELS;
  // tasks = (h chunk, segment, (b, d) slice): at least 4 per workgroup, chunks of >= 8 rows
  const long columns = static_cast<long>(N) * D * (W / 64);
  int h_chunks = 1;
  while (columns * h_chunks < 4L * n_wg && (H + h_chunks) / (h_chunks + 1) >= 8) ++h_chunks;
  const int h_chunk = (H + h_chunks - 1) / h_chunks;
  h_chunks = (H + h_chunk - 1) / h_chunk;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  auto xs = static_cast<const unsigned short*>(x);
  auto ds = static_cast<const unsigned short*>(dy);
#define TRANSOAR_WT_CASE(CI, CO)                                                                         \
  if (ci_n == 8 * CI && co_n == 8 * CO) {                                                                \
    hipLaunchKernelGGL((conv3d_k3_wgrad_tr<CI, CO>), dim3(static_cast<unsigned>(n_wg)), dim3(kWtThreads), 0, st, xs, ds, \
                       partial, N, D, H, W, h_chunks, h_chunk, Cin, ci0, Cout, co0);                  \
    return static_cast<int>(hipGetLastError());                                                          \
  }
  TRANSOAR_WT_CASE(1, 1) TRANSOAR_WT_CASE(1, 2) TRANSOAR_WT_CASE(1, 3) TRANSOAR_WT_CASE(1, 4)
  TRANSOAR_WT_CASE(2, 1) TRANSOAR_WT_CASE(2, 2) TRANSOAR_WT_CASE(2, 3) TRANSOAR_WT_CASE(2, 4)
  TRANSOAR_WT_CASE(3, 1) TRANSOAR_WT_CASE(3, 2) TRANSOAR_WT_CASE(3, 3) TRANSOAR_WT_CASE(3, 4)
  TRANSOAR_WT_CASE(4, 1) TRANSOAR_WT_CASE(4, 2) TRANSOAR_WT_CASE(4, 3) TRANSOAR_WT_CASE(4, 4)
#undef TRANSOAR_WT_CASE
  return TRANSOAR_CONV_ERR_CHANNELS;
}

extern "C" int transoar_conv3d_c1_wgrad(const void* x, const void* dy, float* partial, int n_partial, int N, int D, int H,
                                        int W, int Cout, void* hip_stream) {
  if (!x || !dy || !partial) return TRANSOAR_CONV_ERR_NULL;
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 || n_partial <= 0) return TRANSOAR_CONV_ERR_DIM;
  if (Cout > 32 || (W & 15)) return TRANSOAR_CONV_ERR_CHANNELS;
  const long n_rows = static_cast<long>(N) * D * H;
  const int rows_per_wave = static_cast<int>((n_rows + static_cast<long>(n_partial) * 4 - 1) / (static_cast<long>(n_partial) * 4));
  hipLaunchKernelGGL(conv3d_c1_wgrad, dim3(static_cast<unsigned>(n_partial)), dim3(256), 0,
                     static_cast<hipStream_t>(hip_stream), static_cast<const unsigned short*>(x),
                     static_cast<const unsigned short*>(dy), partial, N, D, H, W, Cout, n_rows, rows_per_wave);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_conv3d_c1_wgrad_tr(const void* x, const void* dy, float* partial, int n_partial, int N, int D, int H,
                                           int W, int Cout, void* hip_stream) {
  if (!x || !dy || !partial) return TRANSOAR_CONV_ERR_NULL;
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 || n_partial <= 0) return TRANSOAR_CONV_ERR_DIM;
  if (Cout > 32 || (Cout & 7) || (W & 63) || W > 256) return TRANSOAR_CONV_ERR_CHANNELS;
  const long n_rows = static_cast<long>(N) * D * H;
  const long xb = n_rows * W * 2, dyb = n_rows * W * Cout * 2;
  if (dyb >= 0x7ffffff0L) return TRANSOAR_CONV_ERR_DIM;
  const int rows_per_wg = static_cast<int>((n_rows + n_partial - 1) / n_partial);
#define TRANSOAR_C1TR(CO8)                                                                                                          \
  hipLaunchKernelGGL(conv3d_c1_wgrad_tr<CO8>, dim3(static_cast<unsigned>(n_partial)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), \
                     static_cast<const unsigned short*>(x), static_cast<const unsigned short*>(dy), partial, N, D, H, W, Cout, n_rows,   \
                     rows_per_wg, static_cast<unsigned>(xb), static_cast<unsigned>(dyb))
  switch (Cout / 8) {
    case 1: TRANSOAR_C1TR(1); break;
    case 2: TRANSOAR_C1TR(2); break;
    case 3: TRANSOAR_C1TR(3); break;
    default: TRANSOAR_C1TR(4); break;
  }
#undef TRANSOAR_C1TR
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_conv3d_c1_forward(const void* x, const float* w, void* y, int N, int D, int H, int W,
                                          int Cout, void* hip_stream) {
  if (!x || !w || !y) return TRANSOAR_CONV_ERR_NULL;
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0) return TRANSOAR_CONV_ERR_DIM;
  if ((Cout & 7) || Cout > transoar::kC1MaxCout) return TRANSOAR_CONV_ERR_CHANNELS;
  const long n_vox = static_cast<long>(N) * D * H * W;
  hipLaunchKernelGGL(conv3d_c1_fwd, dim3(static_cast<unsigned>((n_vox + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(hip_stream), static_cast<const unsigned short*>(x), w,
                     static_cast<unsigned short*>(y), N, D, H, W, Cout, n_vox);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_layout_bf16(const void* in, void* out, int N, long V, int C, int to_channels_first,
                                    void* hip_stream) {
  if (!in || !out) return TRANSOAR_CONV_ERR_NULL;
  if (N <= 0 || V <= 0 || C <= 0) return TRANSOAR_CONV_ERR_DIM;
  if (C & 7) return TRANSOAR_CONV_ERR_CHANNELS;
  const dim3 grid(static_cast<unsigned>((V + 255) / 256), static_cast<unsigned>(N));
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  if (to_channels_first)
    hipLaunchKernelGGL(layout_vc_to_cv, grid, dim3(256), 0, st, static_cast<const unsigned short*>(in),
                       static_cast<unsigned short*>(out), V, C);
  else
    hipLaunchKernelGGL(layout_cv_to_vc, grid, dim3(256), 0, st, static_cast<const unsigned short*>(in),
                       static_cast<unsigned short*>(out), V, C);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_conv3d_abi_version(void) { return 5; }
