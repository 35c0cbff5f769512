// Shared device helpers of the register-stationary, tile-streaming MFMA kernels (gfx950): csrc/attn.hip (fused masked
// cross-attention, Swin window attention) and csrc/gemm_stream.hip (the K = 384 / N = 384 token GEMMs).
//
// Common structure: 256 threads = 4 waves; the stationary operand of a wave's 32 rows lives in registers as MFMA B
// fragments, the streamed operand comes through LDS in tiles of 32 rows x 768 bytes (384 bf16 channels) by LDS-DMA,
// one LDS image serving both the ds_read_b128 fragment reads and the transposing ds_read_b64_tr_b16 reads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace transoar {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int kC = 384;                      // channels of a token (the folded attention's head dimension)
constexpr int kRowBytes = kC * 2;            // 768
constexpr int kKS = kC / 16;                 // 24 MFMA K steps over the channels
constexpr int kCT = kC / 32;                 // 12 channel tiles of an accumulator
constexpr int kTile = 32 * kRowBytes;        // 24 576 bytes: 32 rows
constexpr int kMaxRowTiles = 16;             // rows of a group <= 512 (key-stationary kernel: row statistics of the whole group in LDS)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ int tile_swz(int n) { return ((n & 3) << 2) | ((n >> 2) & 3); }
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
// (bf16(b) << 16) | bf16(a), round to nearest even
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
// v_exp_f32 as it is (exp2f() adds a denormal-range rescue: three more instructions per value); arguments here are <= ~0
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float other_half(float v) { return __shfl_xor(v, 32, 64); }

// One 32-row tile of a dense (rows, C) bf16 matrix, global -> LDS by DMA (buffer_load_dwordx4 ... lds: 16 bytes per
// lane, the LDS side is lane-linear).  rs: descriptor of the whole matrix, tile_byte: byte offset of the tile's row 0
// (wave-uniform, < 2^32).  Rows past the end of the matrix read as zeros (hardware range check); rows of the next
// group are finite data -- whoever consumes such rows masks their contribution.
// Wave w issues pieces [6 w, 6 w + 6) of 1 KiB; lane i of piece j fills LDS bytes (6 w + j) * 1024 + 16 i.
//
// The loads are inline assembly on purpose: issued through __builtin_amdgcn_raw_ptr_buffer_load_lds, hipcc puts
// `s_waitcnt vmcnt(0)` in front of the first transposing LDS read that follows (it cannot tell the DMA's LDS
// destination from the read's source), i.e. the next tile's DMA would only overlap half of a tile's MFMA work.
// hipcc neither counts these loads nor waits for them: every tile loop ends with dma_wait() before its barrier, and
// nothing else in the loops is a vector memory operation.  M0 (the LDS destination base) is saved and restored in
// the statement that uses it; the leading s_nop covers the SGPR-write -> VMEM-read hazard of freshly computed operands.
template <int WAVES = 4>      // waves of the workgroup sharing the tile: 24 / WAVES pieces each
__device__ __forceinline__ void dma_tile(__amdgpu_buffer_rsrc_t rs, unsigned tile_byte, unsigned char* lds_tile, int wave, int lane) {
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>((lds_void*)lds_tile));
  const unsigned soff = __builtin_amdgcn_readfirstlane(tile_byte);
  constexpr int kPer = 24 / WAVES;
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const int piece = kPer * wave + j;
    const int o = piece * 1024 + 16 * lane;
    const int n = ((o >> 8) * 171) >> 9;                 // o / 768 for o < 24 576
    const int q = (o - n * kRowBytes) >> 4;               // physical 16-byte piece inside the row
    const int p = q ^ tile_swz(n);                        // the logical piece that lives there
    const int voff = n * kRowBytes + p * 16;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + piece * 1024);
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
}
// every DMA of this wave has landed (then a barrier, then the reads)
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t matrix_rsrc(const unsigned short* base, long rows) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(base), 0, static_cast<int>(rows * kRowBytes), 0x00020000);
}

// Per-lane LDS byte offsets of the two fragment reads inside a tile, reduced to 16 registers: everything else about
// a read (which tile of the ring stage, K step, channel tile, row half) is an immediate offset of the ds_read.
// (Written as one address expression per read, hipcc hoists ~70 loop-invariant addresses out of the tile loop and
// spills a hundred registers around it.)
//   rows: fragment [32 rows (lane & 31)][16 channels of K step ks]: lane = (row n, half kh) reads the 8 channels
//         16 ks + 8 kh .. + 7 of row n = logical piece 2 ks + kh = 16 (ks >> 3) + (2 (ks & 7) + kh): the swizzle only
//         touches the low four bits -> offset rows[ks & 7] + 256 (ks >> 3)
//   cols: fragment of the TRANSPOSED tile [32 channels of tile ct (lane & 31)][16 rows 16 j + 8 kh .. + 7]: two
//         transposing reads of 4 rows x 64 bytes per 32 lanes (the addressing msda3d_pcm.hpp uses for V): lane (kh, g, r, c)
//         supplies row n = 16 j + 8 kh + r (+ 4), channels 32 ct + 16 g + 4 c .. + 3, i.e. piece 4 ct + 2 g + (c >> 1),
//         byte 8 (c & 1) of it.  swz(n) = 4 r | 2 kh (+ 1 for the second read): with ct = 4 a + b the swizzled piece is
//         16 a + 4 (b ^ r) + ((2 g + (c >> 1)) ^ (2 kh (+ 1))) -> offset cols[b][i] + 256 a + 12288 j
struct FragBase {
  int rows[8];
  int cols[4][2];
};
__device__ __forceinline__ FragBase frag_base(int lane) {
  FragBase fb;
  const int n = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int e = 0; e < 8; ++e) fb.rows[e] = n * kRowBytes + (((2 * e + kh) ^ tile_swz(n)) << 4);
  const int r = (lane & 15) >> 2, g = (lane >> 4) & 1, c = lane & 3;
  const int lp = 2 * g + (c >> 1);
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      fb.cols[b][i] = (8 * kh + r + 4 * i) * kRowBytes + 64 * (b ^ r) + 16 * (lp ^ (2 * kh + i)) + 8 * (c & 1);
  return fb;
}
// move the bases by `delta` bytes (to the other ring stage), in place and opaque to the optimiser: a second set of
// bases costs 16 registers the backward kernels do not have, and visible arithmetic is folded back into ~70 hoisted
// addresses
__device__ __forceinline__ void frag_shift(FragBase& fb, int delta) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    fb.rows[e] += delta;
    asm volatile("" : "+v"(fb.rows[e]));
  }
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fb.cols[b][i] += delta;
      asm volatile("" : "+v"(fb.cols[b][i]));
    }
}
template <int TILE_OFF>
__device__ __forceinline__ s16x8 frag_rows(const unsigned char* lds, const FragBase& fb, int ks) {
  return *reinterpret_cast<const s16x8*>(lds + fb.rows[ks & 7] + (TILE_OFF + 256 * (ks >> 3)));
}
template <int TILE_OFF>
__device__ __forceinline__ s16x8 frag_cols(const unsigned char* lds, const FragBase& fb, int j, int ct) {
  const int imm = TILE_OFF + 256 * (ct >> 2) + 16 * kRowBytes * j;
  const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + fb.cols[ct & 3][0] + imm));
  const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + fb.cols[ct & 3][1] + imm));
  return __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// 16 fp32 values of a 32x32 MFMA result column (entries (r & 3) + 8 (r >> 2) + 4 kh) -> the two B operand fragments
// [k = entry 16 j + 8 kh .. + 7][n = lane & 31] in bf16: v_cvt_pk_bf16_f32 + v_permlane32_swap between the wave halves.
__device__ __forceinline__ void packed_column_to_b_frags(const unsigned (&pk)[8], s16x8 (&frag)[2]) {      // pk[i] = entries 2 i, 2 i + 1
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const auto s0 = __builtin_amdgcn_permlane32_swap(pk[4 * j], pk[4 * j + 2], false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(pk[4 * j + 1], pk[4 * j + 3], false, false);
    frag[j] = __builtin_bit_cast(s16x8, u32x4{s0[0], s1[0], s0[1], s1[1]});
  }
}
__device__ __forceinline__ void column_to_b_frags(const float (&v)[16], s16x8 (&frag)[2]) {
  unsigned pk[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) pk[i] = pack_bf16(v[2 * i], v[2 * i + 1]);
  packed_column_to_b_frags(pk, frag);
}

// the stationary B fragments of one wave: 24 x 16 bytes of row `row` (clamped by the caller) of a (rows, C) matrix
__device__ __forceinline__ void load_row_frags(const unsigned short* __restrict__ row_ptr, int kh, s16x8 (&f)[kKS]) {
#pragma unroll
  for (int ks = 0; ks < kKS; ++ks) f[ks] = *reinterpret_cast<const s16x8*>(row_ptr + 16 * ks + 8 * kh);
}
// "the value is needed HERE": makes hipcc wait for the load that produces it at this point.  Every ordinary load of
// a kernel is pinned like this before its tile loop: a counted wait that hipcc would otherwise place inside the loop
// (vmcnt(N) for a load issued before it) also waits for the loop's DMA loads, which hipcc does not know of.
template <typename T> __device__ __forceinline__ void need(T& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void need_frags(s16x8 (&f)[kKS]) {
#pragma unroll
  for (int ks = 0; ks < kKS; ++ks) need(f[ks]);
}


}  // namespace transoar
