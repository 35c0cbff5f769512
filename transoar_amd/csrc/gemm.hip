// bf16 token GEMM for gfx950:  C[M][N] = A[M][K] . B[N][K]^T (+ bias[N]) (ReLU), fp32 accumulate.
//
// The projections of the deformable-attention refinement (value / output / stacked offsets|attention, FFN
// 384 -> 1024 -> 384; ops/modules/ms_deform_attn.py:109-140, backbones/decoder_blocks.py:157-174) and their
// data gradients: M = 234 000 tokens at batch 2, K and N in {384, 1024}.  Both operands are K-contiguous
// ("NT"), which is nn.Linear's own layout (x (M, K), weight (N, K)): no transposes anywhere.
//
//   * 256 threads = 4 waves (2 x 2), block tile 128 x 128, K step 64; a wave owns 64 x 64 = 2 x 2 MFMA tiles
//     of v_mfma_f32_32x32x16 (bf16 or f16), 16 MFMAs per K step;
//   * operand tiles are [128 rows][64 K] = 128-byte rows in LDS with the 16-byte pieces XOR-swizzled by the
//     row (piece ^ ((row >> 1) & 7)); with row & 7, as in round 2, rows 8 apart shared their four banks and a third of
//     the LDS cycles were conflicts (ds_read_b128 serves lanes {0-3,12-15,20-27} together): the ds_read_b128 of an MFMA fragment (32 rows, one piece each) spreads over all
//     banks; two LDS stages, global -> registers -> LDS staging with the next tile's loads in flight during the
//     MFMAs (one barrier per K step);
//   * the MFMAs are issued with the WEIGHT tile as the A operand, so D = C^T tiles come out [n][m]: a lane
//     holds 4 consecutive n of one token and stores 8 (bf16) or 16 (fp32) bytes;
//   * raw buffer loads: rows past M or N read as zeros, no edge branches in the main loop;
//   * XCD-contiguous block order with the N tiles of one M stripe adjacent (they re-read the same A rows).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/transoar_gemm.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BK = 64;
constexpr int kATileBytes = BM * BK * 2;                  // 16 KiB; the B tile is (64 * WN) rows

template <bool F16>
__device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  return __builtin_bit_cast(unsigned short, static_cast<__bf16>(f));      // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
}
__device__ __forceinline__ unsigned short f32_to_f16(float f) {
  return __builtin_bit_cast(unsigned short, static_cast<_Float16>(f));
}

__device__ __forceinline__ float bf16_lo_f(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi_f(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
// The exact GELU (approximate='none') and its derivative in fp32.  Phi(x) through erfc's rational-exponential form
// (Abramowitz & Stegun 7.1.26: |error| <= 1.5e-7 -- four orders below the bf16 rounding of the result) on the quarter-rate
// rcp / exp2 units: 0.5 erfc(|x| / sqrt 2) = 0.5 t (a1 + t (a2 + ...)) exp(-x^2 / 2), t = 1 / (1 + p |x| / sqrt 2), and the SAME
// exponential is the density's.  (erff + expf of the device library: ~100 instructions per element with two waves per SIMD
// to hide them -- the fused products ran 0.9 ms per step SLOWER than the separate gelu kernels they replaced.)
__device__ __forceinline__ float gelu_phi(float x, float& u) {            // -> Phi(x); u = exp(-x^2 / 2)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.f));
  const float poly = t * __builtin_fmaf(t, __builtin_fmaf(t, __builtin_fmaf(t, __builtin_fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  u = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);            // -0.5 log2(e)
  const float half = 0.5f * poly * u;
  return x < 0.f ? half : 1.f - half;
}
__device__ __forceinline__ float gelu_f(float x) {
  float u;
  return x * gelu_phi(x, u);
}
__device__ __forceinline__ float gelu_grad_f(float dy, float x) {
  float u;
  const float cdf = gelu_phi(x, u);
  return dy * __builtin_fmaf(x, u * 0.39894228040143267794f, cdf);          // Phi + x phi
}
__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {
  return static_cast<unsigned>(f32_to_bf16(a)) | (static_cast<unsigned>(f32_to_bf16(b)) << 16);
}

// LDS-only barrier: __syncthreads() carries a workgroup fence that also drains the outstanding GLOBAL loads
// (s_waitcnt vmcnt(0)), i.e. the prefetch of the next tiles, at every K step.  Here: LDS traffic done, then barrier.
__device__ __forceinline__ void block_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 64 lanes x 16 bytes global -> LDS by DMA (buffer_load_dwordx4 ... lds): lane i lands at LDS byte dst + 16 i (dst, soff
// wave-uniform; the K offset rides in soff).  Inline assembly: hipcc neither counts these loads nor waits for them --
// dma_wait() before the barrier that publishes a stage does (mfma_stream.hpp explains why the builtin is not used).
typedef __attribute__((address_space(3))) void gemm_lds_void;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned dst) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rs), "s"(dst), "s"(soff)
      : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// block id -> (m tile, n tile): each XCD walks a contiguous eighth of the tiles, n fastest
__device__ __forceinline__ long xcd_contiguous(long bid, long n) {
  const long per = (n + 7) >> 3;
  const long swz = (bid & 7) * per + (bid >> 3);
  return swz < n ? swz : -1;
}

// EPI: 0 plain, 1 ReLU, 2 GELU with both results (C = gelu(h), aux = h = A B^T + bias: what the backward needs), 3 the GELU's
// backward on the way out (C = (A B^T) * gelu'(aux)).  2 and 3 work on the bf16-ROUNDED product, in the row-store phase: the
// rounding points of a GEMM followed by torch's gelu / gelu_backward kernels, minus their passes over the hidden tensor.
template <bool F16, bool OUT_F32, int EPI, int KT_STATIC, int WN>
__global__ __launch_bounds__(128 * WN, (WN == 1 || (WN == 3 && EPI != 3)) ? 3 : 2) void gemm_nt_kernel(
    const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, const float* __restrict__ bias,
    void* __restrict__ Cout, unsigned short* __restrict__ aux, int M, int N, int K, int lda, int ldb, int ldc, long n_tiles,
    int tiles_n) {
  constexpr bool RELU = EPI == 1;
  static_assert(EPI < 2 || (!F16 && !OUT_F32), "the GELU epilogues are bf16 in, bf16 out");
  constexpr int BN = 64 * WN, kThreads = 128 * WN;          // WN waves along N (1, 2, 3 or 4), 2 along M
  constexpr int kStageBytes = kATileBytes + BN * BK * 2;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][kStageBytes];      // [stage]: A tile, then B tile
  // Tiles of this workgroup: XCD x = block & 7 owns the contiguous eighth [x per, (x + 1) per) of the tiles (n fastest);
  // its G = gridDim / 8 workgroups take tiles j, j + G, j + 2 G, ... of it.  With the full grid (G = per) that is one
  // tile per workgroup; with the static K loops the host launches ~2 workgroups per CU (PERSIST) and the loads of the
  // NEXT tile's first two K steps are issued under the last two K steps of this one: the 2-3 us of load latency at the
  // head of every tile -- a third of a six-step K loop -- disappear behind the previous tile's MFMAs and epilogue.
  // DPRE (round 5, second half): the run-time K loop does the same with ONE step of look-ahead and a stage index that is
  // carried from tile to tile -- the Swin stages' products have K = 48 .. 192, i.e. one to three K steps per tile, and ran
  // one workgroup per tile: load latency, a handful of MFMAs, epilogue, nothing overlapping.
  constexpr bool DPRE = KT_STATIC == 0 && WN <= 3;
  constexpr bool PERSIST = (KT_STATIC > 0 && (KT_STATIC % 2) == 0) || DPRE;
  const long per = (n_tiles + 7) >> 3;
  const long xend = min((static_cast<long>(blockIdx.x & 7) + 1) * per, n_tiles);
  const long G = gridDim.x >> 3;
  long t = static_cast<long>(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (t >= xend) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;                  // wave's 64 x 64 part of the block tile
  int m0 = static_cast<int>(t / tiles_n) * BM, n0 = static_cast<int>(t % tiles_n) * BN;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(A), 0, static_cast<int>(static_cast<long>(M) * lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(B), 0, static_cast<int>(static_cast<long>(N) * ldb * 2), 0x00020000);

  // staging: thread -> (row, 16-byte piece) x 4 per operand.  The row's pieces past K read as zeros (offset
  // clamp below), rows past M / N are beyond the buffer: zeros as well.
  constexpr int RPP = kThreads / 8;                          // rows covered per pass of the block
  constexpr int NA = (BM + RPP - 1) / RPP, NB = BN / RPP;    // passes per operand tile: (4, 4), (2, 4), (8, 4); WN == 3: 3 (the last one 32 of 48 rows), 4
  const int s_piece = tid & 7, s_row = tid >> 3;             // rows s_row + RPP i
  // Round 5: the tiles go global -> LDS by DMA (round 2-4: through two register sets and 8 ds_write_b128 per thread and K
  // step, 13 cycles each on the CU's store path).  Lane L of wave w fills the PHYSICAL piece s_piece of row s_row + RPP i;
  // the swizzle key ((row >> 1) & 7) is the same for all rows of a thread (RPP is a multiple of 16), so it fetches ONE
  // logical piece of the K step.
  const int l_piece = s_piece ^ ((s_row >> 1) & 7);
  unsigned a_off[NA], b_off[NB];
  // a row past M / N gets an offset beyond any buffer (records < 2^31) that cannot wrap when the K offset is added
  auto tile_offsets = [&](int tm0, int tn0, unsigned (&ao)[NA], unsigned (&bo)[NB]) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int r = s_row + RPP * i;
      ao[i] = (tm0 + r) < M ? static_cast<unsigned>(tm0 + r) * static_cast<unsigned>(lda) * 2u + l_piece * 16u : 0x80000000u;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int r = s_row + RPP * i;
      bo[i] = (tn0 + r) < N ? static_cast<unsigned>(tn0 + r) * static_cast<unsigned>(ldb) * 2u + l_piece * 16u : 0x80000000u;
    }
  };
  tile_offsets(m0, n0, a_off, b_off);
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>((gemm_lds_void*)&lds[0][0]));
  const unsigned dst0 = __builtin_amdgcn_readfirstlane(lds0 + static_cast<unsigned>(wave) * 1024u);
  auto dma_tile = [&](int kt, int stage, const unsigned (&a_off)[NA], const unsigned (&b_off)[NB]) {
    // the K offset rides in the instruction's scalar offset: no per-lane address arithmetic in the loop
    const unsigned kbyte = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(kt * BK * 2));
    const unsigned base = __builtin_amdgcn_readfirstlane(dst0 + static_cast<unsigned>(stage) * kStageBytes);
    const bool in_k = kt * BK + l_piece * 8 < K;             // last, partial K tile: pieces past K read as zeros (K % 8 == 0)
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (BM % RPP == 0 || s_row + RPP * i < BM)            // (WN == 3: waves 4, 5 have no row in the third pass -- wave-uniform)
        dma16(ra, in_k ? a_off[i] : 0x80000000u, kbyte, base + i * (RPP * 128));
#pragma unroll
    for (int i = 0; i < NB; ++i) dma16(rb, in_k ? b_off[i] : 0x80000000u, kbyte, base + kATileBytes + i * (RPP * 128));
  };

  f32x16 acc[2][2];          // [n tile][m tile] of the wave's quadrant, D = [n][m]
  const int KT = (K + BK - 1) / BK;
  const int fr = lane & 31, kg = lane >> 5;
  float4 bv[2][4];
  // fragment offsets inside an operand tile: loop invariant (row * 128 + swizzled piece * 16)
  int fa_off[4][2], fb_off[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rm = wm * 64 + i * 32 + fr, rn = wn * 64 + i * 32 + fr, piece = 2 * ks + kg;
      fa_off[ks][i] = rm * 128 + ((piece ^ ((rm >> 1) & 7)) << 4);
      fb_off[ks][i] = rn * 128 + ((piece ^ ((rn >> 1) & 7)) << 4);
    }
  auto compute = [&](int stage) {
    const unsigned char* ta = &lds[0][0] + stage * kStageBytes;
    const unsigned char* tb = ta + kATileBytes;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const s16x8*>(ta + fa_off[ks][i]);
        fb[i] = *reinterpret_cast<const s16x8*>(tb + fb_off[ks][i]);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mfma<F16>(fb[a], fa[b], acc[a][b]);      // D[n][m] += W[n][k] X[m][k]
    }
  };
  // XPRE: the first K step of the workgroup's NEXT tile is requested under the last K step of this one, into the stage
  // that step leaves free (K steps alternate 0, 1, ..., the count is even: the last one is multiplied out of stage 1); the
  // epilogue's row staging then lives in stage 1 alone.  (128 x 128 tiles only: 8 waves x 8 KiB do not fit one stage.)
  constexpr bool XPRE = PERSIST && WN == 2 && !DPRE;
  if constexpr (XPRE || DPRE) dma_tile(0, 0, a_off, b_off);
  u32x4 hpre[EPI == 3 ? 8 : 1];
  int cur = 0;                            // DPRE: the stage that receives K step 0 of the current tile
  int last = 0;                           //       ... and the one its last K step was multiplied out of (the epilogue stages rows there)
  bool first = true, has_next;
  do {
    if (!first) block_barrier();          // every wave is done with its output turn before the next tile's operands land in LDS
    first = false;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    // the lane's 32 bias values (n = n0 + wn*64 + a*32 + 8q + 4kg + e), requested before the K loop: eight
    // 16-byte loads in flight together instead of 64 dependent scalar loads in the epilogue
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + a * 32 + 8 * q + 4 * kg;
        bv[a][q] = (bias != nullptr && n < N) ? *reinterpret_cast<const float4*>(bias + n) : float4{0.f, 0.f, 0.f, 0.f};
      }
    const long t_next = t + G;
    has_next = PERSIST && t_next < xend;
    const int m0_next = static_cast<int>(t_next / tiles_n) * BM, n0_next = static_cast<int>(t_next % tiles_n) * BN;
    unsigned a_nx[NA], b_nx[NB];
    if constexpr (PERSIST) tile_offsets(m0_next, n0_next, a_nx, b_nx);
  if constexpr (KT_STATIC > 0) {
    // K known at compile time (the token shapes: 384 and 1024): straight-line K loop
    if constexpr (!XPRE) dma_tile(0, 0, a_off, b_off);
    dma_wait();
    block_barrier();
#pragma unroll
    for (int kt = 0; kt < KT_STATIC; ++kt) {
      if (kt + 1 < KT_STATIC) dma_tile(kt + 1, (kt + 1) & 1, a_off, b_off);
      else if (XPRE && has_next) dma_tile(0, 0, a_nx, b_nx);
      compute(kt & 1);
      dma_wait();
      block_barrier();
    }
  } else if constexpr (DPRE) {
    dma_wait();                            // K step 0 of this tile (requested under the previous tile's last step) has landed
    block_barrier();
    if constexpr (EPI == 3) {              // the GELU input of the lane's 8 output pieces: in flight under the K loop and the staging
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int m = m0 + wm * 64 + it * 8 + (lane >> 3), n = n0 + wn * 64 + (lane & 7) * 8;
        hpre[it] = u32x4{0u, 0u, 0u, 0u};
        if (m < M && n + 8 <= N && (ldc & 7) == 0) hpre[it] = *reinterpret_cast<const u32x4*>(aux + static_cast<long>(m) * ldc + n);
      }
    }
    for (int kt = 0; kt < KT; ++kt) {
      const int sg = cur ^ (kt & 1);
      if (kt + 1 < KT) dma_tile(kt + 1, sg ^ 1, a_off, b_off);
      else if (has_next) dma_tile(0, sg ^ 1, a_nx, b_nx);      // stage sg ^ 1: last read two barriers ago (K step / previous epilogue)
      compute(sg);
      if (kt + 1 < KT) dma_wait();          // (the next tile's step is waited for at the top of its own turn, not here)
      block_barrier();
    }
    last = cur ^ ((KT - 1) & 1);
    cur = last ^ 1;
  } else {
    dma_tile(0, 0, a_off, b_off);
    dma_wait();
    block_barrier();
    for (int kt = 0; kt < KT; kt += 2) {
      if (kt + 1 < KT) dma_tile(kt + 1, 1, a_off, b_off);
      compute(0);
      dma_wait();
      block_barrier();
      if (kt + 1 >= KT) break;
      if (kt + 2 < KT) dma_tile(kt + 2, 0, a_off, b_off);
      compute(1);
      dma_wait();
      block_barrier();
    }
  }

  // epilogue.  A lane holds token m = .. + fr, outputs n = .. + 8 q + 4 kg + (0..3): stored straight from the
  // registers that would be 8-byte pieces scattered over 64 rows per instruction.  The wave's 64 x 64 tile goes
  // through its own LDS region instead (row pitch padded by 16 bytes against bank conflicts) and leaves as
  // whole rows: 16 bytes per lane, 8 (4 for fp32) consecutive rows of 128 (256) contiguous bytes per store.
  // Rows of 128 bytes without padding (4 waves x 8 KiB = one stage), the 16-byte pieces XORed with row & 7.
  // WN == 3 (six waves, 128 x 192 tile): 6 x 8 KiB do not fit the one free 40-KiB stage -- the wave turns its result 32 rows
  // at a time through a 4-KiB region instead
  constexpr bool TWO_TURN = WN == 3;
  constexpr int kTurnBytes = TWO_TURN ? 32 * 128 : 64 * 128;
  unsigned char* stage = &lds[0][0] + (DPRE ? last : (XPRE ? 1 : 0)) * kStageBytes + wave * kTurnBytes;
  static_assert(2 * WN * kTurnBytes <= ((XPRE || DPRE) ? 1 : 2) * kStageBytes, "bf16 staging fits");
  // (the main loop's last barrier has passed: every wave is done reading the operand tiles)
  if constexpr (!OUT_F32) {
    auto write_half = [&](int b, int row0) {                          // rows b * 32 + fr of the wave tile -> turn rows row0 + fr
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nl = a * 32 + 8 * q + 4 * kg;                       // column inside the wave tile
          unsigned short h[4];
          const float bq4[4] = {bv[a][q].x, bv[a][q].y, bv[a][q].z, bv[a][q].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[a][b][4 * q + e] + bq4[e];
            if (RELU) v = fmaxf(v, 0.f);
            h[e] = F16 ? f32_to_f16(v) : f32_to_bf16(v);
          }
          *reinterpret_cast<uint2*>(stage + (row0 + fr) * 128 + ((((nl >> 2) >> 1) ^ (fr & 7)) << 4) + ((nl >> 2) & 1) * 8) =
              uint2{static_cast<unsigned>(h[0]) | (static_cast<unsigned>(h[1]) << 16), static_cast<unsigned>(h[2]) | (static_cast<unsigned>(h[3]) << 16)};
        }
    };
    // same wave wrote and reads: LDS operations of a wave are in order
    const int piece = lane & 7, r0 = lane >> 3;
    auto store_rows = [&](int it, int turn_row) {                     // wave-tile row it * 8 + r0, held at turn row turn_row
      const int row = it * 8 + r0;
      const int m = m0 + wm * 64 + row, n = n0 + wn * 64 + piece * 8;
      if (m < M && n < N) {
        u32x4 v = *reinterpret_cast<const u32x4*>(stage + turn_row * 128 + ((piece ^ (turn_row & 7)) << 4));
        const long at = static_cast<long>(m) * ldc + n;
        const bool whole = n + 8 <= N && (ldc & 7) == 0;
        auto put = [&](unsigned short* dst, const u32x4& x) {
          if (whole) *reinterpret_cast<u32x4*>(dst) = x;
          else {
            *reinterpret_cast<uint2*>(dst) = uint2{x[0], x[1]};            // N is a multiple of 4
            if (n + 4 < N) *reinterpret_cast<uint2*>(dst + 4) = uint2{x[2], x[3]};
          }
        };
        if constexpr (EPI == 2) {
          put(aux + at, v);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = pack2_bf16(gelu_f(bf16_lo_f(v[e])), gelu_f(bf16_hi_f(v[e])));
        }
        if constexpr (EPI == 3) {
          u32x4 h{0u, 0u, 0u, 0u};
          if (whole) h = DPRE ? hpre[it] : *reinterpret_cast<const u32x4*>(aux + at);
          else {
            const uint2 h0 = *reinterpret_cast<const uint2*>(aux + at);
            h[0] = h0.x; h[1] = h0.y;
            if (n + 4 < N) { const uint2 h1 = *reinterpret_cast<const uint2*>(aux + at + 4); h[2] = h1.x; h[3] = h1.y; }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = pack2_bf16(gelu_grad_f(bf16_lo_f(v[e]), bf16_lo_f(h[e])), gelu_grad_f(bf16_hi_f(v[e]), bf16_hi_f(h[e])));
        }
        put(static_cast<unsigned short*>(Cout) + at, v);
      }
    };
    if constexpr (!TWO_TURN) {
      write_half(0, 0);
      write_half(1, 32);
#pragma unroll
      for (int it = 0; it < 8; ++it) store_rows(it, it * 8 + r0);
    } else {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        write_half(b, 0);
#pragma unroll
        for (int it4 = 0; it4 < 4; ++it4) store_rows(b * 4 + it4, it4 * 8 + r0);
      }
    }
  } else {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int m = m0 + wm * 64 + b * 32 + fr;
      if (m >= M) continue;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 64 + a * 32 + 8 * q + 4 * kg;
          if (n >= N) continue;
          float v[4];
          const float bq4[4] = {bv[a][q].x, bv[a][q].y, bv[a][q].z, bv[a][q].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[a][b][4 * q + e] + bq4[e];
            if (RELU) v[e] = fmaxf(v[e], 0.f);
          }
          *reinterpret_cast<float4*>(static_cast<float*>(Cout) + static_cast<long>(m) * ldc + n) = float4{v[0], v[1], v[2], v[3]};
        }
    }
  }
    // on to the workgroup's next tile (XPRE: its first K step is already on its way into stage 0)
    if constexpr (PERSIST) {
      t = t_next; m0 = m0_next; n0 = n0_next;
#pragma unroll
      for (int i = 0; i < NA; ++i) a_off[i] = a_nx[i];
#pragma unroll
      for (int i = 0; i < NB; ++i) b_off[i] = b_nx[i];
    }
  } while (has_next);
}

template <bool F16, bool OUT_F32>
int launch(const void* A, const void* B, const float* bias, void* C, void* aux, int M, int N, int K, int lda, int ldb, int ldc,
           int epi, hipStream_t st) {
  // the 128 x 256 tile (8 waves, half the re-reads of the activation rows) measured SLOWER on 234000 x 384 -> 1024
  // (0.41 vs 0.38 ms: one 96-KiB block per CU) -- kept as a template instance, not selected
  const bool wide = false;
  const int BN = wide ? 256 : 128;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const long n_tiles = static_cast<long>(tiles_m) * tiles_n;
  const int kts = epi >= 2 ? 0 : (K == 384 ? 6 : (K == 1024 ? 16 : 0));       // the GELU epilogues: dynamic K loop only
  // static K loops: persistent workgroups, two per CU (TRANSOAR_GEMM_PERSIST_WGS, 0 = one workgroup per tile)
  static const long persist_wgs = [] {
    const char* e = getenv("TRANSOAR_GEMM_PERSIST_WGS");
    return e ? atol(e) : 512L;
  }();
  long per_xcd = (n_tiles + 7) / 8;
  if ((kts > 0 || !wide) && persist_wgs >= 8 && per_xcd > persist_wgs / 8) per_xcd = persist_wgs / 8;
  const dim3 grid(static_cast<unsigned>(per_xcd * 8));
  auto a = static_cast<const unsigned short*>(A);
  auto b = static_cast<const unsigned short*>(B);
  auto x = static_cast<unsigned short*>(aux);
#define TRANSOAR_GEMM_LAUNCH(E, KTS)                                                                                        \
  do {                                                                                                                      \
    if (wide)                                                                                                               \
      hipLaunchKernelGGL((gemm_nt_kernel<F16, OUT_F32, E, KTS, 4>), grid, dim3(512), 0, st, a, b, bias, C, x, M, N, K, lda, ldb, \
                         ldc, n_tiles, tiles_n);                                                                            \
    else                                                                                                                    \
      hipLaunchKernelGGL((gemm_nt_kernel<F16, OUT_F32, E, KTS, 2>), grid, dim3(256), 0, st, a, b, bias, C, x, M, N, K, lda, ldb, \
                         ldc, n_tiles, tiles_n);                                                                            \
  } while (0)
  if constexpr (!F16 && !OUT_F32) {
    static const int narrow_max = [] { const char* e = getenv("TRANSOAR_GEMM_NARROW_MAX"); return e ? atoi(e) : 64; }();
    if (epi == 0 && kts == 0 && N <= narrow_max) {
      // up to 64 output columns (the Swin blocks' 48-wide products): 128 x 64 tiles on two waves -- the 128-wide tile left
      // half of its waves without a column to store and DMA-ed 64 out-of-range weight rows per K step; three workgroups per CU
      const int tn = (N + 63) / 64;
      const long nt = static_cast<long>(tiles_m) * tn;
      long px = (nt + 7) / 8;
      if (persist_wgs >= 8 && px > persist_wgs * 3 / 16) px = persist_wgs * 3 / 16;
      hipLaunchKernelGGL((gemm_nt_kernel<false, false, 0, 0, 1>), dim3(static_cast<unsigned>(px * 8)), dim3(128), 0, st, a, b, bias, C, x, M, N, K,
                         lda, ldb, ldc, nt, tn);
      return static_cast<int>(hipGetLastError());
    }
    static const bool use_192 = [] { const char* e = getenv("TRANSOAR_GEMM_TILE192"); return !(e && e[0] == '0'); }();
    if (use_192 && kts == 0 && N > 128 && N <= 192) {
      // 129 .. 192 output columns (the Swin blocks' 144- and 192-wide products): ONE 128 x 192 tile on six waves instead of a
      // full and a mostly empty 128-wide one
      long px = (static_cast<long>(tiles_m) + 7) / 8;
      if (persist_wgs >= 8 && px > persist_wgs / 8) px = persist_wgs / 8;
      const dim3 g3(static_cast<unsigned>(px * 8));
      const long nt = tiles_m;
      if (epi == 2) hipLaunchKernelGGL((gemm_nt_kernel<false, false, 2, 0, 3>), g3, dim3(384), 0, st, a, b, bias, C, x, M, N, K, lda, ldb, ldc, nt, 1);
      else if (epi == 3) hipLaunchKernelGGL((gemm_nt_kernel<false, false, 3, 0, 3>), g3, dim3(384), 0, st, a, b, bias, C, x, M, N, K, lda, ldb, ldc, nt, 1);
      else if (epi == 1) hipLaunchKernelGGL((gemm_nt_kernel<false, false, 1, 0, 3>), g3, dim3(384), 0, st, a, b, bias, C, x, M, N, K, lda, ldb, ldc, nt, 1);
      else hipLaunchKernelGGL((gemm_nt_kernel<false, false, 0, 0, 3>), g3, dim3(384), 0, st, a, b, bias, C, x, M, N, K, lda, ldb, ldc, nt, 1);
      return static_cast<int>(hipGetLastError());
    }
    if (epi == 2) { TRANSOAR_GEMM_LAUNCH(2, 0); return static_cast<int>(hipGetLastError()); }
    if (epi == 3) { TRANSOAR_GEMM_LAUNCH(3, 0); return static_cast<int>(hipGetLastError()); }
  }
  if (epi >= 2) return TRANSOAR_GEMM_ERR_DTYPE;
  if (epi == 1) {
    if (kts == 6) TRANSOAR_GEMM_LAUNCH(1, 6);
    else if (kts == 16) TRANSOAR_GEMM_LAUNCH(1, 16);
    else TRANSOAR_GEMM_LAUNCH(1, 0);
  } else {
    if (kts == 6) TRANSOAR_GEMM_LAUNCH(0, 6);
    else if (kts == 16) TRANSOAR_GEMM_LAUNCH(0, 16);
    else TRANSOAR_GEMM_LAUNCH(0, 0);
  }
#undef TRANSOAR_GEMM_LAUNCH
  return static_cast<int>(hipGetLastError());
}

}  // namespace

static int check_nt(const void* A, const void* B, const float* bias, const void* C, int M, int N, int K, int lda, int ldb, int ldc) {
  if (!A || !B || !C) return TRANSOAR_GEMM_ERR_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || (K & 7) || (N & 3) || lda < K || ldb < K || ldc < N || (lda & 7) || (ldb & 7) || (ldc & 3))
    return TRANSOAR_GEMM_ERR_DIM;
  if (static_cast<long>(N) * ldb * 2 >= 0x7ffffff0L || static_cast<long>(M) * lda * 2 >= 0x7ffffff0L) return TRANSOAR_GEMM_ERR_DIM;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias)) & 15u) return TRANSOAR_GEMM_ERR_ALIGN;
  return 0;
}

extern "C" int transoar_gemm_nt(const void* A, const void* B, const float* bias, void* C, int M, int N, int K, int lda,
                                int ldb, int ldc, int in_dtype, int out_dtype, int relu, void* hip_stream) {
  if (const int rc = check_nt(A, B, bias, C, M, N, K, lda, ldb, ldc)) return rc;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const bool f16 = in_dtype == TRANSOAR_GEMM_F16;
  const int epi = relu ? 1 : 0;
  if (in_dtype != TRANSOAR_GEMM_BF16 && !f16) return TRANSOAR_GEMM_ERR_DTYPE;
  if (out_dtype == TRANSOAR_GEMM_F32) return f16 ? launch<true, true>(A, B, bias, C, nullptr, M, N, K, lda, ldb, ldc, epi, st)
                                                 : launch<false, true>(A, B, bias, C, nullptr, M, N, K, lda, ldb, ldc, epi, st);
  if (out_dtype != in_dtype) return TRANSOAR_GEMM_ERR_DTYPE;
  return f16 ? launch<true, false>(A, B, bias, C, nullptr, M, N, K, lda, ldb, ldc, epi, st)
             : launch<false, false>(A, B, bias, C, nullptr, M, N, K, lda, ldb, ldc, epi, st);
}

extern "C" int transoar_gemm_nt_gelu(const void* A, const void* B, const float* bias, void* C, void* aux, int M, int N, int K,
                                     int lda, int ldb, int ldc, int mode, void* hip_stream) {
  if (!aux) return TRANSOAR_GEMM_ERR_NULL;
  if (const int rc = check_nt(A, B, bias, C, M, N, K, lda, ldb, ldc)) return rc;
  if (reinterpret_cast<uintptr_t>(aux) & 15u) return TRANSOAR_GEMM_ERR_ALIGN;
  if (mode != TRANSOAR_GEMM_GELU_FORWARD && mode != TRANSOAR_GEMM_GELU_BACKWARD) return TRANSOAR_GEMM_ERR_DIM;
  return launch<false, false>(A, B, bias, C, aux, M, N, K, lda, ldb, ldc, mode == TRANSOAR_GEMM_GELU_FORWARD ? 2 : 3,
                              static_cast<hipStream_t>(hip_stream));
}

extern "C" int transoar_gemm_abi_version(void) { return 4; }
