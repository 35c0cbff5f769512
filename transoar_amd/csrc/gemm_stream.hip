// Token GEMMs with one dimension fixed at 384, register-stationary / tile-streaming form (gfx950).
//
// The refinement block's dense projections (transoar/models/ops/modules/ms_deform_attn.py:109-140: value_proj,
// sampling_offsets | attention_weights, output_proj; transoar/models/backbones/decoder_blocks.py:157-174: linear1 +
// ReLU, linear2) and their data gradients are products  C[M][N] = A[M][K] . B[N][K]^T  over M = 234 000 tokens with
// K = 384 or N = 384.  csrc/gemm.hip's 128 x 128 LDS-tiled kernel spends a third of such a short K loop (6 steps) in
// its prologue and epilogue and lost to hipBLASLt on the FFN shapes (0.38 / 0.27 ms against 0.27 / 0.22).  Here, with
// the machinery of the fused attention kernels (mfma_stream.hpp):
//
//   gemm_k384  (K = 384: forward of every projection and of linear1, data gradient of linear2)
//     a wave keeps ITS 32 tokens' rows of A as 24 MFMA B fragments in registers for the whole kernel; the rows of the
//     weight stream through LDS in tiles of 32 output channels (32 x 768 bytes, LDS-DMA, two-deep ring shared by the
//     four waves); per tile 24 MFMAs produce a complete 32 x 32 output tile C^T[n][token] (bias, ReLU in registers),
//     which leaves through a per-wave LDS turn as whole 128-byte lines.  A is read from HBM exactly once.
//   gemm_n384  (N = 384: forward of linear2, data gradient of linear1)
//     a wave accumulates C^T[384][32 tokens] (192 accumulator registers) over K in chunks of 32: the weight chunk
//     [384 rows][32 k] (24 KB, shared) and the wave's own A chunk [32 tokens][32 k] (2 KB) arrive by LDS-DMA in a
//     five-deep ring (three chunks ahead, counted vmcnt, one bare barrier per chunk); 24 MFMAs per chunk on 12
//     independent accumulators.
//     A is read exactly once, C written once as whole 768-byte rows.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/transoar_gemm.h"
#include "mfma_stream.hpp"

namespace transoar {

// ---------------------------------------------------------------------------
// K = 384
// ---------------------------------------------------------------------------
constexpr int kOutPitch = 144;                 // bytes per token row of a wave's output turn: 64 channels + 16 (bank spread)
constexpr int kOutBytes = 32 * kOutPitch;      // 4 608 per wave
// Eight waves (256 tokens) per workgroup, one workgroup per CU: the weight tiles are an LDS-DMA stream, and that path
// moves ~6 TB/s over the whole chip (MI355X_MICROARCH.md, "ldsdma-fill"; measured here: 5-5.7 TB/s), so the kernel
// is bound by (tokens / tokens per workgroup) x |W|.  Two 4-wave workgroups per CU streamed W twice per CU.
constexpr int kK384Waves = 8;

// seeded dropout of csrc/tokens.hip (keep_pair): element pair p = (flat element index) / 2 is kept where the 16-bit halves
// of hash32(p * 0x9e3779b9 + seed) are below thr16 (low half: the even element)
__device__ __forceinline__ unsigned drop_hash(unsigned pair, unsigned seed) {
  unsigned x = pair * 0x9e3779b9u + seed;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// one 1-KiB LDS-DMA piece: lane i fills LDS bytes [16 i, 16 i + 16) of lds_piece from byte voff (per lane) + soff (uniform)
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rs, unsigned soff_in, int voff, unsigned char* lds_piece) {
  const unsigned dst = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>((lds_void*)lds_piece)));
  const unsigned soff = __builtin_amdgcn_readfirstlane(soff_in);
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
// MASK: C = mask(y) ? (A B^T) * drop_scale : 0 with mask(y) = y > 0 read from `gate` (M, N) bf16 -- the gradient of
// dropout(relu(.)) applied to the product that feeds it (the data gradient of the FFN's second layer): gate is the FFN's
// saved hidden tensor, whose positive entries are exactly the kept, active ones.
template <bool RELU, bool DROP, bool MASK = false>
__global__ __launch_bounds__(kK384Waves * 64) void gemm_k384_kernel(
    const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, const float* __restrict__ bias,
    unsigned short* __restrict__ C, int M, int N, const int* __restrict__ drop_seed, unsigned thr16, float drop_scale,
    const unsigned short* __restrict__ gate = nullptr) {
  // (MASK: + 4 KiB per wave for the gate values of a tile pair, fetched by LDS-DMA with the weight tile: read in the store
  // loop itself they were a global-load round trip per 8 tokens with nothing to hide it -- 0.44 ms against 0.25 for the
  // plain product)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * kTile + kK384Waves * kOutBytes + (MASK ? kK384Waves * 4096 + 1024 : 0)];
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const int kh = lane >> 5;
  FragBase fs = frag_base(lane);
  const long m0w = static_cast<long>(blockIdx.x) * (32 * kK384Waves) + wave * 32;           // first token of this wave
  const long m = m0w + (lane & 31);
  s16x8 xf[kKS];
  load_row_frags(A + (m < M ? m : M - 1) * kC, kh, xf);
  need_frags(xf);
  const __amdgpu_buffer_rsrc_t brs = matrix_rsrc(B, N);
  unsigned seed = 0u;
  if (DROP) {
    seed = static_cast<unsigned>(*drop_seed);
    need(seed);
  }
  const unsigned pair_row = static_cast<unsigned>(m) * static_cast<unsigned>(N >> 1);        // pair index of (token m, channel 0)
  unsigned char* outb = lds + 2 * kTile + wave * kOutBytes;
  const int n_tiles = N >> 5;

  constexpr int kGateOff = ((2 * kTile + kK384Waves * kOutBytes + 1023) / 1024) * 1024;
  unsigned char* gate_lds = lds + kGateOff + wave * 4096;
  __amdgpu_buffer_rsrc_t grs = brs;
  int gate_voff = 0;
  if (MASK) {
    grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(gate), 0, static_cast<int>(static_cast<long>(M) * N * 2), 0x00020000);
    gate_voff = static_cast<int>((static_cast<unsigned>(m0w) + static_cast<unsigned>(lane >> 3)) * static_cast<unsigned>(N) * 2u + static_cast<unsigned>(lane & 7) * 16u);
  }
  dma_tile<kK384Waves>(brs, 0u, lds, wave, lane);
  dma_wait();
  __syncthreads();
  for (int t = 0; t < n_tiles; ++t) {
    const int st = t & 1;
    if (MASK && (t & 1)) {          // the gate's 32 tokens x 128 bytes of this tile pair: 4 pieces of 8 tokens
#pragma unroll
      for (int it = 0; it < 4; ++it)
        dma_piece(grs, static_cast<unsigned>(it * 8) * static_cast<unsigned>(N) * 2u + 64u * static_cast<unsigned>(t - 1), gate_voff, gate_lds + it * 1024);
    }
    if (t + 1 < n_tiles) dma_tile<kK384Waves>(brs, static_cast<unsigned>(t + 1) * kTile, lds + (st ^ 1) * kTile, wave, lane);
    // C^T[n][token] of the tile: two accumulators, weight fragments fetched six K steps ahead
    f32x16 c0, c1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    {
      s16x8 fw[2][6];
#pragma unroll
      for (int e = 0; e < 6; ++e) fw[0][e] = frag_rows<0>(lds, fs, e);
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        if (gq + 1 < 4) {
#pragma unroll
          for (int e = 0; e < 6; ++e) fw[(gq + 1) & 1][e] = frag_rows<0>(lds, fs, 6 * (gq + 1) + e);
        }
#pragma unroll
        for (int e = 0; e < 6; e += 2) {
          c0 = mfma(fw[gq & 1][e], xf[6 * gq + e], c0);
          c1 = mfma(fw[gq & 1][e + 1], xf[6 * gq + e + 1], c1);
        }
      }
    }
    // bias (wave-uniform address: scalar loads), ReLU, bf16 -> the wave's output turn: entry r = channel
    // 32 t + (r & 3) + 8 (r >> 2) + 4 kh of token lane & 31
    {
      const float* bt = bias + 32 * t;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float b = 0.f;
          if (bias != nullptr) b = kh ? bt[8 * qd + 4 + e] : bt[8 * qd + e];
          v[e] = c0[4 * qd + e] + c1[4 * qd + e] + b;
          if (RELU) v[e] = fmaxf(v[e], 0.f);
          if (MASK) v[e] *= drop_scale;                 // scaled in fp32, rounded once; zeroed below where the gate is not positive
        }
        if (DROP) {                               // channels 32 t + 8 qd + 4 kh .. + 3 = two element pairs
          const unsigned p0 = pair_row + static_cast<unsigned>(16 * t + 4 * qd + 2 * kh);
          const unsigned h0 = drop_hash(p0, seed), h1 = drop_hash(p0 + 1u, seed);
          v[0] = (h0 & 0xffffu) < thr16 ? v[0] * drop_scale : 0.f;
          v[1] = (h0 >> 16) < thr16 ? v[1] * drop_scale : 0.f;
          v[2] = (h1 & 0xffffu) < thr16 ? v[2] * drop_scale : 0.f;
          v[3] = (h1 >> 16) < thr16 ? v[3] * drop_scale : 0.f;
        }
        *reinterpret_cast<u32x2*>(outb + (lane & 31) * kOutPitch + 2 * (32 * (t & 1) + 8 * qd + 4 * kh)) =
            u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
      }
    }
    frag_shift(fs, st ? -kTile : kTile);
    dma_wait();
    __syncthreads();
    if (t & 1) {
      // two tiles = 64 channels = one 128-byte line per token: 8 tokens per store instruction
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), piece = lane & 7;
        u32x4 v = *reinterpret_cast<const u32x4*>(outb + row * kOutPitch + piece * 16);
        if (m0w + row < M) {
          if (MASK) {
            // bf16 > 0  <=>  sign bit clear and not zero (the gate holds no NaN: it is a ReLU output)
            const u32x4 y = *reinterpret_cast<const u32x4*>(gate_lds + it * 1024 + lane * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned lo = (y[e] & 0xffffu) - 1u < 0x7fffu ? 0x0000ffffu : 0u;
              const unsigned hi = (y[e] >> 16) - 1u < 0x7fffu ? 0xffff0000u : 0u;
              v[e] &= lo | hi;
            }
          }
          *reinterpret_cast<u32x4*>(C + (m0w + row) * N + 32 * (t - 1) + 8 * piece) = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// N = 384
// ---------------------------------------------------------------------------
constexpr int kChunkK = 32;                                  // K elements per chunk
constexpr int kWChunk = kC * kChunkK * 2;                    // 24 576: [384 rows][64 bytes]
constexpr int kAChunk = 32 * kChunkK * 2;                    // 2 048 per wave: [32 tokens][64 bytes]
constexpr int kStage = kWChunk + 4 * kAChunk;                // 32 768
constexpr int kRing = 5;                                     // 5 x 32 KiB = all of the CU's LDS: one workgroup per CU
constexpr int kRowPitchOut = kRowBytes + 16;                 // 784: output turn [32 tokens][384] per wave

// [rows][64 bytes] chunk, 16 rows per 1-KiB DMA piece: lane = (row in piece, 16-byte piece q); the four pieces of a
// row are XORed with (row >> 2) & 3 (conflict-free ds_read_b128 of 16 rows)
__device__ __forceinline__ void dma_chunk_piece(__amdgpu_buffer_rsrc_t rs, unsigned base_byte, int row0, unsigned row_bytes,
                                                unsigned char* lds_piece, int lane) {
  const int row = row0 + (lane >> 2), q = lane & 3;
  const int voff = row * static_cast<int>(row_bytes) + ((q ^ ((row >> 2) & 3)) << 4);
  const unsigned dst = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>((lds_void*)lds_piece)));
  const unsigned soff = __builtin_amdgcn_readfirstlane(base_byte);
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

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_n384_kernel(
    const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, const float* __restrict__ bias,
    unsigned short* __restrict__ C, int M, int K) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[kRing * kStage];
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const int kh = lane >> 5;
  const long m0w = static_cast<long>(blockIdx.x) * 128 + wave * 32;
  const unsigned row_bytes = static_cast<unsigned>(K) * 2u;
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A), 0, static_cast<int>(static_cast<long>(M) * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(B), 0, static_cast<int>(static_cast<long>(kC) * K * 2), 0x00020000);
  const unsigned a_base = static_cast<unsigned>(m0w) * row_bytes;            // < 2^32: checked by the host
  // fragment offsets inside a chunk: row (lane & 31) (+ 32 ct for the weight), K step ks of the chunk
  int fb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) fb[ks] = (lane & 31) * 64 + (((2 * ks + kh) ^ (((lane & 31) >> 2) & 3)) << 4);
  f32x16 acc[kCT];
#pragma unroll
  for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

  const int n_chunks = K / kChunkK;
  auto issue = [&](int c) {                    // 8 DMA instructions per wave: 6 pieces of the weight chunk, 2 of its own A chunk
    unsigned char* stg = lds + (c % kRing) * kStage;
#pragma unroll
    for (int j = 0; j < 6; ++j) dma_chunk_piece(brs, static_cast<unsigned>(c) * 64u, 16 * (6 * wave + j), row_bytes, stg + (6 * wave + j) * 1024, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) dma_chunk_piece(ars, a_base + static_cast<unsigned>(c) * 64u, 16 * j, row_bytes, stg + kWChunk + wave * kAChunk + j * 1024, lane);
  };
  // the waits count DMA instructions (8 per chunk and wave, in issue order): chunk c has landed when at most the
  // instructions of the chunks issued after it are outstanding
  issue(0);
  if (n_chunks > 1) issue(1);
  if (n_chunks > 2) issue(2);
  for (int c = 0; c < n_chunks; ++c) {
    if (c + 3 < n_chunks) {
      issue(c + 3);
      asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    } else if (c + 2 < n_chunks) {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else if (c + 1 < n_chunks) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // every wave's pieces of chunk c have landed (a bare barrier: no vmcnt(0) drain).  ONE barrier per chunk: the stage
    // read here is overwritten by chunk c + 5, issued at the top of iteration c + 2 by a wave that has passed barrier
    // c + 1, which every wave reaches only after its reads of this iteration
    __builtin_amdgcn_s_barrier();
    int so = (c % kRing) * kStage;
    asm volatile("" : "+s"(so));
    const unsigned char* wt = lds + so;
    const unsigned char* at = wt + kWChunk + wave * kAChunk;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const s16x8 bfr = *reinterpret_cast<const s16x8*>(at + fb[ks]);
#pragma unroll
      for (int ct = 0; ct < kCT; ++ct)
        acc[ct] = mfma(*reinterpret_cast<const s16x8*>(wt + fb[ks] + ct * 2048), bfr, acc[ct]);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                 // the ring becomes the output turn
  // ---- C^T[n][token] + bias -> the wave's LDS turn [token][384] -> whole rows
  unsigned char* outb = lds + wave * (32 * kRowPitchOut);
#pragma unroll
  for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int n = 32 * ct + 8 * qd + 4 * kh;
      float4 b4{0.f, 0.f, 0.f, 0.f};
      if (bias != nullptr) b4 = *reinterpret_cast<const float4*>(bias + n);
      *reinterpret_cast<u32x2*>(outb + (lane & 31) * kRowPitchOut + 2 * n) =
          u32x2{pack_bf16(acc[ct][4 * qd] + b4.x, acc[ct][4 * qd + 1] + b4.y), pack_bf16(acc[ct][4 * qd + 2] + b4.z, acc[ct][4 * qd + 3] + b4.w)};
    }
#pragma unroll
  for (int it = 0; it < 24; ++it) {
    const int e = it * 64 + lane;               // 16-byte piece e of the wave's 32 rows x 48 pieces
    const int row = (e * 683) >> 15;            // e / 48 for e < 1536
    const int pc = e - row * 48;
    if (m0w + row < M)
      *reinterpret_cast<u32x4*>(C + (m0w + row) * kC + 8 * pc) = *reinterpret_cast<const u32x4*>(outb + row * kRowPitchOut + pc * 16);
  }
}

// ---------------------------------------------------------------------------
// weight gradient with 384 columns on one side: Out = A^T B over the token axis
// ---------------------------------------------------------------------------
// A (T, Na) and B (T, 384) bf16, tokens outermost (the layer's input and the gradient of its output, or the other way
// round).  A workgroup owns 32 NT columns of A (NT = 4 or 8) x all 384 columns of B for one chunk of tokens; wave w
// holds the accumulators of B's columns [96 w, 96 w + 96): NT x 3 MFMA tiles (96 / 192 fp32 registers).  Tokens stream
// through LDS in stages of 32 (the B tile in the 32 x 768-byte layout of mfma_stream.hpp, the A tile as [32 tokens]
// [64 NT bytes], its 64-byte windows XORed with (token & 3) so that the transposing reads of four token rows hit
// distinct banks), four-stage ring, two stages ahead, one bare barrier per stage.  Both operands of an MFMA come from
// transposing reads (channel = lane & 31, eight tokens per lane): NT + 3 fragments feed 3 NT MFMAs per 16 tokens.
// Each workgroup writes its fp32 partial (its chunk of tokens); wgrad384_reduce sums the chunks.
// Block order: the NT-tiles of one token chunk sit on the same XCD (block % 8) next to each other, so the chunk's
// B tiles come from that XCD's L2 for all but the first of them.
template <int NT> struct WgTile {
  static constexpr int kRing = NT == 4 ? 5 : 4;         // 5 x 32 KiB / 4 x 40 KiB = all of the CU's LDS
  static constexpr int kAhead = kRing - 2;              // stages in flight beyond the one being read
  static constexpr int kPitch = 64 * NT;                // bytes per token row of the A tile
  static constexpr int kABytes = 32 * kPitch;
  static constexpr int kStage = kTile + kABytes;        // 32 / 40 KiB
  static constexpr int kPerWave = kABytes / 4096;       // 1-KiB DMA pieces of the A tile per wave
  static constexpr int kIssue = 6 + kPerWave;           // DMA instructions per stage and wave
};

__device__ __forceinline__ s16x8 tr_frag(const unsigned char* p0, const unsigned char* p1) {
  const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
  const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
  return __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int NT, bool TR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wgrad384_kernel(
    const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, float* __restrict__ part,
    int T, int Na, int chunk_len, int n_chunks, int n_tiles, float* __restrict__ bias_part, int bias_side) {
  using W = WgTile<NT>;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[W::kRing * W::kStage];
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const int bx = blockIdx.x & 7, by = blockIdx.x >> 3;
  const int tile = by % n_tiles, chunk = (by / n_tiles) * 8 + bx;
  if (chunk >= n_chunks) return;
  const int n0 = tile * 32 * NT;
  const int t0 = chunk * chunk_len;
  const int t_end = min(T, t0 + chunk_len);
  const int n_stages = (t_end - t0 + 31) >> 5;
  const unsigned lda_bytes = static_cast<unsigned>(Na) * 2u;
  // the descriptors end at the chunk's last token: the rows of the last stage past it read as zeros
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A), 0, static_cast<int>(static_cast<unsigned>(t_end) * lda_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(B), 0, static_cast<int>(static_cast<unsigned>(t_end) * static_cast<unsigned>(kRowBytes)), 0x00020000);
  const int kh = lane >> 5, r = (lane & 15) >> 2, g = (lane >> 4) & 1, c = lane & 3;

  // DMA source offsets of this wave's pieces of the A tile (lane-linear LDS image, windows swizzled at the source)
  int avoff[W::kPerWave];
#pragma unroll
  for (int j = 0; j < W::kPerWave; ++j) {
    const int o = (W::kPerWave * wave + j) * 1024 + 16 * lane;
    const int row = o / W::kPitch, q = (o % W::kPitch) >> 4;
    avoff[j] = row * static_cast<int>(lda_bytes) + ((q ^ ((row & 3) << 2)) << 4);
  }
  // fragment bases (mfma_stream.hpp, FragBase::cols, for this wave's three channel tiles of B; the same scheme on the A tile)
  int abase[4], bbase[3][2];
#pragma unroll
  for (int b = 0; b < 4; ++b) abase[b] = kTile + (8 * kh + r) * W::kPitch + 64 * (b ^ r) + 32 * g + 8 * c;
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) {
    const int ct = 3 * wave + jj;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      bbase[jj][i] = (8 * kh + r + 4 * i) * kRowBytes + 64 * ((ct & 3) ^ r) + 16 * ((2 * g + (c >> 1)) ^ (2 * kh + i)) + 8 * (c & 1) + 256 * (ct >> 2);
  }
  f32x16 acc[NT][3];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int jj = 0; jj < 3; ++jj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;

  // Column sums of one operand on the side (the layer's bias gradient: sum over the tokens of dY): the fragments are
  // in registers anyway (sixteen VALU instructions per fragment, beside the matrix pipeline).  bias_side 1: A's columns (every workgroup
  // its own 32 NT columns), 2: B's 384 columns (the workgroups of column tile 0, every wave its 96).
  auto frag_sum = [&](s16x8 f, float acc) {
    const u32x4 w = __builtin_bit_cast(u32x4, f);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += bf16_lo(w[e]) + bf16_hi(w[e]);
    return acc;
  };
  // (A's column tiles are shared out among the four waves, NT / 4 each: on one wave the sums cost 0.36 ms per step at
  // the barrier of every stage)
  const bool sum_a = bias_side == 1, sum_b = bias_side == 2 && tile == 0;
  float asum[NT], bsum[3];
#pragma unroll
  for (int i = 0; i < NT; ++i) asum[i] = 0.f;
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) bsum[jj] = 0.f;
  const unsigned a_soff0 = static_cast<unsigned>(t0) * lda_bytes + static_cast<unsigned>(n0) * 2u;
  auto issue = [&](int s) {
    unsigned char* stg = lds + (s % W::kRing) * W::kStage;
    dma_tile(brs, static_cast<unsigned>(t0 + 32 * s) * static_cast<unsigned>(kRowBytes), stg, wave, lane);
#pragma unroll
    for (int j = 0; j < W::kPerWave; ++j)
      dma_piece(ars, a_soff0 + static_cast<unsigned>(32 * s) * lda_bytes, avoff[j], stg + kTile + (W::kPerWave * wave + j) * 1024);
  };
  issue(0);
  if (n_stages > 1) issue(1);
  if (W::kAhead > 2 && n_stages > 2) issue(2);
  for (int s = 0; s < n_stages; ++s) {
    // the waits count DMA instructions (kIssue per stage and wave, in issue order)
    if (s + W::kAhead < n_stages) {
      issue(s + W::kAhead);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W::kAhead * W::kIssue) : "memory");
    } else if (W::kAhead > 2 && s + 2 < n_stages) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * W::kIssue) : "memory");
    } else if (s + 1 < n_stages) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W::kIssue) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // one barrier per stage: the stage read here is overwritten by stage s + kRing, issued at the top of iteration s + 2
    // by a wave that has passed barrier s + 1, which every wave reaches only after its reads of this iteration
    __builtin_amdgcn_s_barrier();
    int so = (s % W::kRing) * W::kStage;
    asm volatile("" : "+s"(so));
    const unsigned char* st = lds + so;
    // (NT = 8 not unrolled: with both halves of the stage in one block hipcc keeps ~44 fragments in flight and spills
    // the accumulators; NT = 4 unrolled: the second half's reads overlap the first half's MFMAs)
#pragma unroll(NT == 4 ? 2 : 1)
    for (int j = 0; j < 2; ++j) {
      const unsigned char* bt = st + 16 * kRowBytes * j;
      const unsigned char* at = st + 16 * W::kPitch * j;
      s16x8 bf[3];
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) bf[jj] = tr_frag(bt + bbase[jj][0], bt + bbase[jj][1]);
      if (sum_b) {
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) bsum[jj] = frag_sum(bf[jj], bsum[jj]);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int imm = 256 * (i >> 2);
        const s16x8 af = tr_frag(at + abase[i & 3] + imm, at + abase[i & 3] + imm + 4 * W::kPitch);
        if (sum_a && i / (NT / 4) == wave) asum[i] = frag_sum(af, asum[i]);
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
          const s16x8 m_a = TR ? bf[jj] : af, m_b = TR ? af : bf[jj];
          if (3 * i + jj >= 16) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i][jj]) : "v"(m_a), "v"(m_b));
          else acc[i][jj] = mfma(m_a, m_b, acc[i][jj]);
        }
      }
    }
  }
  // NT = 8: 24 accumulator tiles = 384 registers, 16 of them in the 256 AGPRs; hipcc only emits the AGPR form of an
  // MFMA at one wave per SIMD and, left alone, swaps the other eight tiles through AGPRs around every MFMA (320 moves
  // per 24 MFMAs): those eight are inline assembly on VGPR accumulators.  The assembler statements are invisible to
  // the hazard recogniser, hence the wait states before the results are read.
  if (NT == 8) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  if (sum_a) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      if (i / (NT / 4) != wave) continue;
      const float v = asum[i] + other_half(asum[i]);
      if (lane < 32) bias_part[static_cast<long>(chunk) * Na + n0 + 32 * i + lane] = v;
    }
  }
  if (sum_b) {
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const float v = bsum[jj] + other_half(bsum[jj]);
      if (lane < 32) bias_part[static_cast<long>(chunk) * kC + 96 * wave + 32 * jj + lane] = v;
    }
  }
  // ---- the chunk's partial: (Na, 384) row-major, or (384, Na) when TR
  float* out = part + static_cast<long>(chunk) * Na * kC;
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int jj = 0; jj < 3; ++jj)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int d_row = (e & 3) + 8 * (e >> 2) + 4 * kh, d_col = lane & 31;
        if (TR) out[static_cast<long>(96 * wave + 32 * jj + d_row) * Na + n0 + 32 * i + d_col] = acc[i][jj][e];
        else out[static_cast<long>(n0 + 32 * i + d_row) * kC + 96 * wave + 32 * jj + d_col] = acc[i][jj][e];
      }
}

// out[e] = sum over the chunks of part[chunk][e]: 32 float4 elements per workgroup, eight lanes of chunks each (a
// thread sums chunks g, g + 8, ...), then across the eight through LDS in a fixed order
__global__ __launch_bounds__(256) void wgrad384_reduce(const float4* __restrict__ part, float4* __restrict__ out, int n4, int n_chunks) {
  __shared__ float4 sh[8][32];
  const int e = blockIdx.x * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
  float4 s{0.f, 0.f, 0.f, 0.f};
  if (e < n4) {
#pragma unroll 4
    for (int ch = g; ch < n_chunks; ch += 8) {
      const float4 v = part[static_cast<long>(ch) * n4 + e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  sh[g][threadIdx.x & 31] = s;
  __syncthreads();
  if (g == 0 && e < n4) {
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const float4 v = sh[k][threadIdx.x];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    out[e] = s;
  }
}

__global__ __launch_bounds__(256) void wgrad384_bias_reduce(const float* __restrict__ part, float* __restrict__ out, int cols, int n_chunks) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int ch = 0; ch < n_chunks; ++ch) s += part[static_cast<long>(ch) * cols + c];
  out[c] = s;
}

}  // namespace transoar

using namespace transoar;

static unsigned drop_threshold(float keep_prob) {          // csrc/tokens.hip: keep_threshold
  const float t = keep_prob * 65536.f + 0.5f;
  return t <= 0.f ? 0u : (t >= 65535.f ? 65535u : static_cast<unsigned>(t));
}

extern "C" int transoar_gemm_k384_drop(const void* A, const void* B, const float* bias, void* C, int M, int N, int relu,
                                       const int* drop_seed, float keep_prob, float keep_scale, void* hip_stream);

extern "C" int transoar_gemm_k384(const void* A, const void* B, const float* bias, void* C, int M, int N, int relu, void* hip_stream) {
  return transoar_gemm_k384_drop(A, B, bias, C, M, N, relu, nullptr, 1.f, 1.f, hip_stream);
}

extern "C" int transoar_gemm_k384_drop(const void* A, const void* B, const float* bias, void* C, int M, int N, int relu,
                                       const int* drop_seed, float keep_prob, float keep_scale, void* hip_stream) {
  if (!A || !B || !C) return TRANSOAR_GEMM_ERR_NULL;
  if (M <= 0 || N <= 0 || (N & 63)) return TRANSOAR_GEMM_ERR_DIM;
  if (static_cast<long>(N) * kRowBytes >= 0x7ffffff0L || static_cast<long>(M) * N * 2 >= (1L << 40)) return TRANSOAR_GEMM_ERR_DIM;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias)) & 15u)
    return TRANSOAR_GEMM_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const dim3 grid(static_cast<unsigned>((M + 32 * kK384Waves - 1) / (32 * kK384Waves)));
  auto a = static_cast<const unsigned short*>(A);
  auto b = static_cast<const unsigned short*>(B);
  auto c = static_cast<unsigned short*>(C);
  if (static_cast<long>(M) * (N >> 1) >= (1L << 32)) return TRANSOAR_GEMM_ERR_DIM;          // 32-bit pair indices of the dropout hash
  const unsigned thr = drop_threshold(keep_prob);
  if (drop_seed != nullptr) {
    if (relu) hipLaunchKernelGGL((gemm_k384_kernel<true, true>), grid, dim3(64 * kK384Waves), 0, st, a, b, bias, c, M, N, drop_seed, thr, keep_scale);
    else hipLaunchKernelGGL((gemm_k384_kernel<false, true>), grid, dim3(64 * kK384Waves), 0, st, a, b, bias, c, M, N, drop_seed, thr, keep_scale);
  } else {
    if (relu) hipLaunchKernelGGL((gemm_k384_kernel<true, false>), grid, dim3(64 * kK384Waves), 0, st, a, b, bias, c, M, N, drop_seed, thr, keep_scale);
    else hipLaunchKernelGGL((gemm_k384_kernel<false, false>), grid, dim3(64 * kK384Waves), 0, st, a, b, bias, c, M, N, drop_seed, thr, keep_scale);
  }
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_gemm_k384_gate(const void* A, const void* B, const void* gate, void* C, int M, int N, float scale, void* hip_stream) {
  if (!A || !B || !C || !gate) return TRANSOAR_GEMM_ERR_NULL;
  if (M <= 0 || N <= 0 || (N & 63)) return TRANSOAR_GEMM_ERR_DIM;
  // the gate is read through a buffer descriptor of M * N * 2 bytes and 32-bit byte offsets (a 32-row tile may start past M)
  if (static_cast<long>(N) * kRowBytes >= 0x7ffffff0L || (static_cast<long>(M) + 32) * N * 2 >= 0xffffffffL) return TRANSOAR_GEMM_ERR_DIM;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(gate)) & 15u)
    return TRANSOAR_GEMM_ERR_ALIGN;
  hipLaunchKernelGGL((gemm_k384_kernel<false, false, true>), dim3(static_cast<unsigned>((M + 32 * kK384Waves - 1) / (32 * kK384Waves))),
                     dim3(64 * kK384Waves), 0, static_cast<hipStream_t>(hip_stream), static_cast<const unsigned short*>(A),
                     static_cast<const unsigned short*>(B), nullptr, static_cast<unsigned short*>(C), M, N, nullptr, 0u, scale,
                     static_cast<const unsigned short*>(gate));
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_gemm_n384(const void* A, const void* B, const float* bias, void* C, int M, int K, void* hip_stream) {
  if (!A || !B || !C) return TRANSOAR_GEMM_ERR_NULL;
  if (M <= 0 || K <= 0 || (K & 31)) return TRANSOAR_GEMM_ERR_DIM;
  if ((static_cast<long>(M) + 128) * K * 2 >= 0xffffffffL || static_cast<long>(kC) * K * 2 >= 0x7ffffff0L) return TRANSOAR_GEMM_ERR_DIM;     // 32-bit byte offsets into A
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias)) & 15u)
    return TRANSOAR_GEMM_ERR_ALIGN;
  hipLaunchKernelGGL(gemm_n384_kernel, dim3(static_cast<unsigned>((M + 127) / 128)), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                     static_cast<const unsigned short*>(A), static_cast<const unsigned short*>(B), bias, static_cast<unsigned short*>(C), M, K);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_gemm_wgrad384_chunks(int T, int Na) {
  if (T <= 0 || Na <= 0 || (Na & 127)) return 0;
  const int tiles = (Na & 255) ? Na / 128 : Na / 256;
  int chunks = 256 / tiles;
  if (chunks < 1) chunks = 1;
  const int len = ((T + chunks - 1) / chunks + 31) & ~31;
  return (T + len - 1) / len;
}

extern "C" int transoar_gemm_wgrad384_bias(const void* A, const void* B, float* part, float* out, int T, int Na, int transpose_out,
                                           int chunks, float* bias_part, float* bias_out, int bias_side, void* hip_stream);

extern "C" int transoar_gemm_wgrad384(const void* A, const void* B, float* part, float* out, int T, int Na, int transpose_out,
                                      int chunks, void* hip_stream) {
  return transoar_gemm_wgrad384_bias(A, B, part, out, T, Na, transpose_out, chunks, nullptr, nullptr, 0, hip_stream);
}

extern "C" int transoar_gemm_wgrad384_bias(const void* A, const void* B, float* part, float* out, int T, int Na, int transpose_out,
                                           int chunks, float* bias_part, float* bias_out, int bias_side, void* hip_stream) {
  if (!A || !B || !part || !out) return TRANSOAR_GEMM_ERR_NULL;
  if (bias_side < 0 || bias_side > 2 || (bias_side != 0 && (!bias_part || !bias_out))) return bias_side < 0 || bias_side > 2 ? TRANSOAR_GEMM_ERR_DIM : TRANSOAR_GEMM_ERR_NULL;
  if (T <= 0 || Na <= 0 || (Na & 127) || chunks <= 0) return TRANSOAR_GEMM_ERR_DIM;
  if (static_cast<long>(T) * Na * 2 >= 0x7ffffff0L || static_cast<long>(T) * kRowBytes >= 0x7ffffff0L) return TRANSOAR_GEMM_ERR_DIM;   // 32-bit byte offsets
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(out)) & 15u)
    return TRANSOAR_GEMM_ERR_ALIGN;
  const bool wide = (Na & 255) == 0;
  const int tiles = wide ? Na / 256 : Na / 128;
  const int len = ((T + chunks - 1) / chunks + 31) & ~31;
  if (static_cast<long>(len) * chunks < T) return TRANSOAR_GEMM_ERR_DIM;
  const int n_chunks = (T + len - 1) / len;
  if (n_chunks != chunks) return TRANSOAR_GEMM_ERR_DIM;             // the caller sized `part` with transoar_gemm_wgrad384_chunks
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const dim3 grid(static_cast<unsigned>(8 * tiles * ((n_chunks + 7) / 8)));
  auto a = static_cast<const unsigned short*>(A);
  auto b = static_cast<const unsigned short*>(B);
  if (wide) {
    if (transpose_out) hipLaunchKernelGGL((wgrad384_kernel<8, true>), grid, dim3(256), 0, st, a, b, part, T, Na, len, n_chunks, tiles, bias_part, bias_side);
    else hipLaunchKernelGGL((wgrad384_kernel<8, false>), grid, dim3(256), 0, st, a, b, part, T, Na, len, n_chunks, tiles, bias_part, bias_side);
  } else {
    if (transpose_out) hipLaunchKernelGGL((wgrad384_kernel<4, true>), grid, dim3(256), 0, st, a, b, part, T, Na, len, n_chunks, tiles, bias_part, bias_side);
    else hipLaunchKernelGGL((wgrad384_kernel<4, false>), grid, dim3(256), 0, st, a, b, part, T, Na, len, n_chunks, tiles, bias_part, bias_side);
  }
  const int n4 = Na * kC / 4;
  hipLaunchKernelGGL(wgrad384_reduce, dim3((n4 + 31) / 32), dim3(256), 0, st, reinterpret_cast<const float4*>(part), reinterpret_cast<float4*>(out), n4, n_chunks);
  if (bias_side != 0) {
    const int cols = bias_side == 1 ? Na : kC;
    hipLaunchKernelGGL(wgrad384_bias_reduce, dim3((cols + 255) / 256), dim3(256), 0, st, bias_part, bias_out, cols, n_chunks);
  }
  return static_cast<int>(hipGetLastError());
}
