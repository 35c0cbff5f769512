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

// LDS-only barrier: __syncthreads() carries a workgroup fence that also drains the outstanding GLOBAL loads
// (s_waitcnt vmcnt(0)), i.e. the prefetch of the next tiles, at every K step.  Here: LDS traffic done, then barrier.
__device__ __forceinline__ void block_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// block id -> (m tile, n tile): each XCD walks a contiguous eighth of the tiles, n fastest
__device__ __forceinline__ long xcd_contiguous(long bid, long n) {
  const long per = (n + 7) >> 3;
  const long swz = (bid & 7) * per + (bid >> 3);
  return swz < n ? swz : -1;
}

template <bool F16, bool OUT_F32, bool RELU, int KT_STATIC, int WN>
__global__ __launch_bounds__(128 * WN, WN == 2 ? 2 : 2) void gemm_nt_kernel(
    const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, const float* __restrict__ bias,
    void* __restrict__ Cout, int M, int N, int K, int lda, int ldb, int ldc, long n_tiles, int tiles_n) {
  constexpr int BN = 64 * WN, kThreads = 128 * WN;          // WN waves along N (2 or 4), 2 along M
  constexpr int kStageBytes = kATileBytes + BN * BK * 2;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][kStageBytes];      // [stage]: A tile, then B tile
  // Tiles of this workgroup: XCD x = block & 7 owns the contiguous eighth [x per, (x + 1) per) of the tiles (n fastest);
  // its G = gridDim / 8 workgroups take tiles j, j + G, j + 2 G, ... of it.  With the full grid (G = per) that is one
  // tile per workgroup; with the static K loops the host launches ~2 workgroups per CU (PERSIST) and the loads of the
  // NEXT tile's first two K steps are issued under the last two K steps of this one: the 2-3 us of load latency at the
  // head of every tile -- a third of a six-step K loop -- disappear behind the previous tile's MFMAs and epilogue.
  constexpr bool PERSIST = KT_STATIC > 0 && (KT_STATIC % 2) == 0;
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
  constexpr int NA = BM / RPP, NB = BN / RPP;                // passes per operand tile: (4, 4) or (2, 4)
  const int s_piece = tid & 7, s_row = tid >> 3;             // rows s_row + RPP i
  unsigned a_off[NA], b_off[NB];
  // a row past M / N gets an offset beyond any buffer (records < 2^31) that cannot wrap when the K offset is added
  auto tile_offsets = [&](int tm0, int tn0, unsigned (&ao)[NA], unsigned (&bo)[NB]) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int r = s_row + RPP * i;
      ao[i] = (tm0 + r) < M ? static_cast<unsigned>(tm0 + r) * static_cast<unsigned>(lda) * 2u + s_piece * 16u : 0x80000000u;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int r = s_row + RPP * i;
      bo[i] = (tn0 + r) < N ? static_cast<unsigned>(tn0 + r) * static_cast<unsigned>(ldb) * 2u + s_piece * 16u : 0x80000000u;
    }
  };
  tile_offsets(m0, n0, a_off, b_off);
  // two register sets: tile kt+2 is requested while tile kt is multiplied and tile kt+1 waits in the other set
  // (one tile ahead leaves the HBM latency exposed: a K step is only ~500 cycles of MFMA per wave)
  u32x4 ra0[NA], rb0[NB], ra1[NA], rb1[NB];
  auto load_tile = [&](int kt, const unsigned (&a_off)[NA], const unsigned (&b_off)[NB], u32x4 (&ra_regs)[NA], u32x4 (&rb_regs)[NB]) {
    // the K offset rides in the instruction's scalar offset: no per-lane address arithmetic in the loop
    const int kbyte = kt * BK * 2;
    if ((kt + 1) * BK <= K) {
#pragma unroll
      for (int i = 0; i < NA; ++i) ra_regs[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, a_off[i], kbyte, 0);
#pragma unroll
      for (int i = 0; i < NB; ++i) rb_regs[i] = __builtin_amdgcn_raw_buffer_load_b128(rb, b_off[i], kbyte, 0);
    } else {                                                 // last, partial K tile: pieces past K read as zeros
      const bool in_k = kt * BK + s_piece * 8 < K;           // K is a multiple of 8 (checked on the host)
#pragma unroll
      for (int i = 0; i < NA; ++i) ra_regs[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, in_k ? a_off[i] : 0x80000000u, kbyte, 0);
#pragma unroll
      for (int i = 0; i < NB; ++i) rb_regs[i] = __builtin_amdgcn_raw_buffer_load_b128(rb, in_k ? b_off[i] : 0x80000000u, kbyte, 0);
    }
  };
  auto store_tile = [&](int stage, const u32x4 (&ra_regs)[NA], const u32x4 (&rb_regs)[NB]) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int r = s_row + RPP * i;
      *reinterpret_cast<u32x4*>(&lds[stage][r * 128 + ((s_piece ^ ((r >> 1) & 7)) << 4)]) = ra_regs[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int r = s_row + RPP * i;
      *reinterpret_cast<u32x4*>(&lds[stage][kATileBytes + r * 128 + ((s_piece ^ ((r >> 1) & 7)) << 4)]) = rb_regs[i];
    }
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
    const unsigned char* ta = lds[stage];
    const unsigned char* tb = lds[stage] + kATileBytes;
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
  if constexpr (PERSIST) {
    load_tile(0, a_off, b_off, ra0, rb0);
    load_tile(1, a_off, b_off, ra1, rb1);
  }
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
    // K known at compile time (the token shapes: 384 and 1024): the K loop is straight-line code, so the
    // compiler's wait counts are exact -- the ds_write of tile kt+1 waits for ITS loads only and leaves the
    // loads of tile kt+2 in flight (across the loop back-edge of the generic form it drains them)
    if constexpr (!PERSIST) {
      load_tile(0, a_off, b_off, ra0, rb0);
      if (KT_STATIC > 1) load_tile(1, a_off, b_off, ra1, rb1);
    }
    store_tile(0, ra0, rb0);
    block_barrier();
#pragma unroll
    for (int kt = 0; kt < KT_STATIC; ++kt) {
      if (kt & 1) {
        if (kt + 2 < KT_STATIC) load_tile(kt + 2, a_off, b_off, ra1, rb1);
        else if (PERSIST && has_next) load_tile(kt + 2 - KT_STATIC, a_nx, b_nx, ra1, rb1);
        compute(1);
        if (kt + 1 < KT_STATIC) store_tile(0, ra0, rb0);
      } else {
        if (kt + 2 < KT_STATIC) load_tile(kt + 2, a_off, b_off, ra0, rb0);
        else if (PERSIST && has_next) load_tile(kt + 2 - KT_STATIC, a_nx, b_nx, ra0, rb0);
        compute(0);
        if (kt + 1 < KT_STATIC) store_tile(1, ra1, rb1);
      }
      block_barrier();
    }
  } else {
  load_tile(0, a_off, b_off, ra0, rb0);
  if (KT > 1) load_tile(1, a_off, b_off, ra1, rb1);
  store_tile(0, ra0, rb0);
  block_barrier();
  for (int kt = 0; kt < KT; kt += 2) {
    // even step: LDS stage 0 holds tile kt; set 0 is free, set 1 holds tile kt+1
    if (kt + 2 < KT) load_tile(kt + 2, a_off, b_off, ra0, rb0);
    compute(0);
    if (kt + 1 < KT) store_tile(1, ra1, rb1);
    block_barrier();
    if (kt + 1 >= KT) break;
    // odd step: stage 1 holds tile kt+1; set 1 is free, set 0 holds tile kt+2
    if (kt + 3 < KT) load_tile(kt + 3, a_off, b_off, ra1, rb1);
    compute(1);
    if (kt + 2 < KT) store_tile(0, ra0, rb0);
    block_barrier();
  }
  }

  // epilogue.  A lane holds token m = .. + fr, outputs n = .. + 8 q + 4 kg + (0..3): stored straight from the
  // registers that would be 8-byte pieces scattered over 64 rows per instruction.  The wave's 64 x 64 tile goes
  // through its own LDS region instead (row pitch padded by 16 bytes against bank conflicts) and leaves as
  // whole rows: 16 bytes per lane, 8 (4 for fp32) consecutive rows of 128 (256) contiguous bytes per store.
  constexpr int ELT = OUT_F32 ? 4 : 2;
  constexpr int PITCH = 64 * ELT + 16;
  unsigned char* stage = &lds[0][0] + wave * (64 * PITCH);
  static_assert(2 * WN * 64 * (64 * 2 + 16) <= 2 * kStageBytes, "bf16 staging fits");
  // (the main loop's last barrier has passed: every wave is done reading the operand tiles)
  if constexpr (!OUT_F32) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
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
          *reinterpret_cast<uint2*>(stage + (b * 32 + fr) * PITCH + nl * 2) =
              uint2{static_cast<unsigned>(h[0]) | (static_cast<unsigned>(h[1]) << 16), static_cast<unsigned>(h[2]) | (static_cast<unsigned>(h[3]) << 16)};
        }
    // same wave wrote and reads: LDS operations of a wave are in order
    const int piece = lane & 7, r0 = lane >> 3;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + r0;
      const int m = m0 + wm * 64 + row, n = n0 + wn * 64 + piece * 8;
      if (m < M && n < N) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(stage + row * PITCH + piece * 16);
        unsigned short* dst = static_cast<unsigned short*>(Cout) + static_cast<long>(m) * ldc + n;
        if (n + 8 <= N && (ldc & 7) == 0) *reinterpret_cast<u32x4*>(dst) = v;
        else {
          *reinterpret_cast<uint2*>(dst) = uint2{v[0], v[1]};            // N is a multiple of 4
          if (n + 4 < N) *reinterpret_cast<uint2*>(dst + 4) = uint2{v[2], v[3]};
        }
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
    // on to the workgroup's next tile (its first two K steps are already in the register sets)
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
int launch(const void* A, const void* B, const float* bias, void* C, int M, int N, int K, int lda, int ldb, int ldc, int relu,
           hipStream_t st) {
  // the 128 x 256 tile (8 waves, half the re-reads of the activation rows) measured SLOWER on 234000 x 384 -> 1024
  // (0.41 vs 0.38 ms: one 96-KiB block per CU) -- kept as a template instance, not selected
  const bool wide = false;
  const int BN = wide ? 256 : 128;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const long n_tiles = static_cast<long>(tiles_m) * tiles_n;
  const int kts = K == 384 ? 6 : (K == 1024 ? 16 : 0);
  // static K loops: persistent workgroups, two per CU (TRANSOAR_GEMM_PERSIST_WGS, 0 = one workgroup per tile)
  static const long persist_wgs = [] {
    const char* e = getenv("TRANSOAR_GEMM_PERSIST_WGS");
    return e ? atol(e) : 512L;
  }();
  long per_xcd = (n_tiles + 7) / 8;
  if (kts > 0 && persist_wgs >= 8 && per_xcd > persist_wgs / 8) per_xcd = persist_wgs / 8;
  const dim3 grid(static_cast<unsigned>(per_xcd * 8));
  auto a = static_cast<const unsigned short*>(A);
  auto b = static_cast<const unsigned short*>(B);
#define TRANSOAR_GEMM_LAUNCH(R, KTS)                                                                                        \
  do {                                                                                                                      \
    if (wide)                                                                                                               \
      hipLaunchKernelGGL((gemm_nt_kernel<F16, OUT_F32, R, KTS, 4>), grid, dim3(512), 0, st, a, b, bias, C, M, N, K, lda, ldb, \
                         ldc, n_tiles, tiles_n);                                                                            \
    else                                                                                                                    \
      hipLaunchKernelGGL((gemm_nt_kernel<F16, OUT_F32, R, KTS, 2>), grid, dim3(256), 0, st, a, b, bias, C, M, N, K, lda, ldb, \
                         ldc, n_tiles, tiles_n);                                                                            \
  } while (0)
  if (relu) {
    if (kts == 6) TRANSOAR_GEMM_LAUNCH(true, 6);
    else if (kts == 16) TRANSOAR_GEMM_LAUNCH(true, 16);
    else TRANSOAR_GEMM_LAUNCH(true, 0);
  } else {
    if (kts == 6) TRANSOAR_GEMM_LAUNCH(false, 6);
    else if (kts == 16) TRANSOAR_GEMM_LAUNCH(false, 16);
    else TRANSOAR_GEMM_LAUNCH(false, 0);
  }
#undef TRANSOAR_GEMM_LAUNCH
  return static_cast<int>(hipGetLastError());
}

}  // namespace

extern "C" int transoar_gemm_nt(const void* A, const void* B, const float* bias, void* C, int M, int N, int K, int lda,
                                int ldb, int ldc, int in_dtype, int out_dtype, int relu, void* hip_stream) {
  if (!A || !B || !C) return TRANSOAR_GEMM_ERR_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || (K & 7) || (N & 3) || lda < K || ldb < K || ldc < N || (lda & 7) || (ldb & 7) || (ldc & 3))
    return TRANSOAR_GEMM_ERR_DIM;
  if (static_cast<long>(N) * ldb * 2 >= 0x7ffffff0L || static_cast<long>(M) * lda * 2 >= 0x7ffffff0L) return TRANSOAR_GEMM_ERR_DIM;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias)) & 15u) return TRANSOAR_GEMM_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const bool f16 = in_dtype == TRANSOAR_GEMM_F16;
  if (in_dtype != TRANSOAR_GEMM_BF16 && !f16) return TRANSOAR_GEMM_ERR_DTYPE;
  if (out_dtype == TRANSOAR_GEMM_F32) return f16 ? launch<true, true>(A, B, bias, C, M, N, K, lda, ldb, ldc, relu, st)
                                                 : launch<false, true>(A, B, bias, C, M, N, K, lda, ldb, ldc, relu, st);
  if (out_dtype != in_dtype) return TRANSOAR_GEMM_ERR_DTYPE;
  return f16 ? launch<true, false>(A, B, bias, C, M, N, K, lda, ldb, ldc, relu, st)
             : launch<false, false>(A, B, bias, C, M, N, K, lda, ldb, ldc, relu, st);
}

extern "C" int transoar_gemm_abi_version(void) { return 3; }
