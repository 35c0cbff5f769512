// 3x3x3 convolutions of the deep encoder stages and the FPN as an LDS-tiled implicit GEMM on the matrix cores (gfx950).
//
// Reference layers: EncoderCnnBlock (backbones/encoder_blocks.py:28-51: Conv3d(3, stride s, pad 1, no bias), stages
// 1-5 at 48..768 channels) and the FPN `out` convolutions (backbones/attn_fpn.py:65-73,126: Conv3d(3, pad 1) + bias,
// 96..384 -> 384).  conv3d.hip covers the full-resolution layers (<= 32 channels, halo tile in LDS); this file covers
// everything from 48 channels up, forward and both gradients, so that no 3^3 convolution is left to MIOpen:
//
//   transoar_conv3d_igemm   Y[m][n] = sum_{tap, c} X[src(m, tap)][c] * Wk[tap][n][c] (+ bias[n])
//       rows m = voxels of a "row space" (MD, MH, MW) per batch element, src(m, tap) = stride * m + delta(tap) in the
//       source map, out-of-range sources read as zeros (the padding), the result row goes to voxel
//       out_stride * m + out_parity of the output map.  One kernel, three uses:
//         forward          row space = output map, stride 1 or 2, taps = all 27 with delta = t - 1
//         data gradient    of a stride-1 layer: the same with the flipped, in/out-swapped filter (host packs it)
//         data gradient    of a stride-2 layer: ONE launch over the eight parity classes (pd, ph, pw) of the dx
//                          voxels -- a dx voxel of even coordinate sees tap 1 of dy voxel x/2, one of odd coordinate
//                          taps 0 and 2 of dy voxels (x+1)/2 and (x-1)/2: row space = the class's voxels, 1/2/4/8
//                          taps, source = dy at stride 1, output rows strided by 2 into dx
//       Launches with few tiles split K: every split stores its fp32 tile into its own partial map (no atomics) and
//       conv3d_finish_kernel sums the maps, adds the bias and casts.
//   transoar_conv3d_wgrad   dW[tap][co][ci] = sum_m dY[m][co] * X[src(m, tap)][ci]  (msda-style "TN" GEMM: both
//       operands have the contraction axis (voxels) as their slow axis; tiles are staged K-major and read with the
//       transposing ds_read_b64_tr_b16), split over voxel chunks; each chunk stores its fp32 tile into its own
//       partial map and conv3d_wgrad_reduce_kernel sums them into (Cout, Cin, 27) through an LDS turn.  With taps = 1
//       and no shift it is the weight gradient of a token projection (dW = dY^T X, decoder_blocks.py:157-174).
//
// GEMM machinery = gemm.hip's: 256 threads = 2 x 2 waves, block tile 128 x 128, K step 64, v_mfma_f32_32x32x16_bf16,
// operand tiles as 128-byte rows in LDS with 16-byte pieces XOR-swizzled by (row >> 1) & 7, two LDS stages + two register sets, raw
// buffer loads whose out-of-range lanes read zeros.  K is the flattened (tap, channel) axis in pieces of 8
// channels: a K step of 64 may straddle two taps (Cin = 48, 96 ...), every thread resolves its own piece's tap.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/transoar_convgemm.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BK = 64;
constexpr int kTile = BM * BK * 2;               // 16 KiB per operand tile

__device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  return __builtin_bit_cast(unsigned short, static_cast<__bf16>(f));      // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
}
// 64 lanes x 16 bytes global -> LDS by DMA (buffer_load_dwordx4 ... lds): lane i lands at LDS byte dst + 16 i (dst wave-uniform).
// Inline assembly: hipcc neither counts these loads nor waits for them -- dma_wait() before the barrier that publishes a
// stage does (mfma_stream.hpp explains why the builtin is not used).
typedef __attribute__((address_space(3))) void conv_lds_void;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned dst) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rs), "s"(dst)
      : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// LDS-only barrier (see gemm.hip): does not drain the global loads in flight
__device__ __forceinline__ void block_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ long xcd_contiguous(long bid, long n) {
  const long per = (n + 7) >> 3;
  const long swz = (bid & 7) * per + (bid >> 3);
  return swz < n ? swz : -1;
}

// per-dimension tap list, packed by the host: bits [1:0] = count (1..3), then per entry e: bits [2+4e +: 2] = delta + 1,
// bits [4+4e +: 2] = filter index t
__device__ __forceinline__ int taps_count(unsigned p) { return p & 3; }
__device__ __forceinline__ int taps_delta(unsigned p, int e) { return static_cast<int>((p >> (2 + 4 * e)) & 3) - 1; }
__device__ __forceinline__ int taps_t(unsigned p, int e) { return (p >> (4 + 4 * e)) & 3; }

struct ConvGeom {
  int N, SD, SH, SW;            // source map
  int MD, MH, MW;               // row space (per batch element)
  int OD, OH, OW;               // output map
  int src_stride, out_stride, opd, oph, opw;
  int Cin, Cout;                // contraction channels per tap, output channels
};

// ---------------------------------------------------------------------------------------------------------------
// Y = conv(X, Wk) as implicit GEMM.  Block tile 128 rows x (64 TN) output channels, 2 x 2 waves.
//   split: the K steps are divided among `split` workgroups per tile; with y32 != NULL every workgroup stores its fp32
//          partial tile to y32[split index][output row][n] (plain 16-byte stores; an empty K range stores zeros) and
//          transoar_conv3d_finish sums the partials, adds the bias and casts; else (split == 1) bf16 rows go to Y.
//   classes != 0: data gradient of a stride-2 layer, all eight parity classes of the dx voxels in ONE launch: the
//          output map (OD, OH, OW) is dx, the source map dy; class (pd, ph, pw) has the rows (OD - pd + 1) / 2 x ...,
//          1, 2, 4 or 8 taps (per axis, parity 0: tap 1 of dy voxel m; parity 1: taps 0, 2 of dy voxels m + 1, m) and
//          writes dx voxels 2 m + parity.  Tiles are numbered class by class, the eight-tap class first.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kInvalid = 0x40000000;            // 2^30: beyond every filter pack (check_geom)
constexpr unsigned kTapsParity0 = 1u | (1u << 2) | (1u << 4);                                   // {(0, 1)}
constexpr unsigned kTapsParity1 = 2u | (2u << 2) | (0u << 4) | (1u << 6) | (2u << 8);           // {(+1, 0), (0, 2)}

template <int TN>
__global__ __launch_bounds__(256, 2) void conv3d_igemm_kernel(
    const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wk, const float* __restrict__ bias,
    unsigned short* __restrict__ Y, float* __restrict__ y32, ConvGeom g, unsigned tp_d, unsigned tp_h, unsigned tp_w,
    int split, long n_tiles, int tiles_n, int classes, unsigned x_bytes, unsigned w_bytes) {
  constexpr int BNT = 64 * TN, NB = BNT / 32;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][kTile + BNT * BK * 2];
  __shared__ int2 tap_tab[32];                 // {source offset, filter offset} per tap-list entry; entries >= ntaps: out of range
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long tile_id = xcd_contiguous(blockIdx.x, n_tiles * split);
  if (tile_id < 0) return;
  const int ks = static_cast<int>(tile_id / n_tiles);              // split index: slowest, so the tiles of one split are adjacent
  long t = tile_id - static_cast<long>(ks) * n_tiles;
  if (classes) {
    // which parity class does tile t belong to?  (wave-uniform; classes in the order 7 .. 0)
    for (int c = 7; c >= 0; --c) {
      const int pd = (c >> 2) & 1, ph = (c >> 1) & 1, pw = c & 1;
      const int md = (g.OD - pd + 1) >> 1, mh = (g.OH - ph + 1) >> 1, mw = (g.OW - pw + 1) >> 1;
      const long tl = (static_cast<long>(g.N) * md * mh * mw + BM - 1) / BM * tiles_n;
      if (t < tl || c == 0) {
        g.MD = md; g.MH = mh; g.MW = mw; g.opd = pd; g.oph = ph; g.opw = pw;
        tp_d = pd ? kTapsParity1 : kTapsParity0; tp_h = ph ? kTapsParity1 : kTapsParity0; tp_w = pw ? kTapsParity1 : kTapsParity0;
        break;
      }
      t -= tl;
    }
    if (g.MD * g.MH * g.MW == 0) return;
  }
  const int tm = static_cast<int>(t / tiles_n), tn = static_cast<int>(t % tiles_n);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = tm * BM, n0 = tn * BNT;
  const int M = g.N * g.MD * g.MH * g.MW;
  const int nd = taps_count(tp_d), nh = taps_count(tp_h), nw = taps_count(tp_w);
  const int ntaps = nd * nh * nw;
  if (tid < 32) {
    int2 e{0, kInvalid};                                          // a piece past the end of K reads zeros from both operands
    if (tid < ntaps) {
      const int ed = tid / (nh * nw), r = tid - ed * nh * nw, eh = r / nw, ew = r - eh * nw;
      e.x = ((taps_delta(tp_d, ed) * g.SH + taps_delta(tp_h, eh)) * g.SW + taps_delta(tp_w, ew)) * g.Cin * 2;
      e.y = ((taps_t(tp_d, ed) * 3 + taps_t(tp_h, eh)) * 3 + taps_t(tp_w, ew)) * g.Cout * g.Cin * 2;
    }
    tap_tab[tid] = e;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(X), 0, static_cast<int>(x_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Wk), 0, static_cast<int>(w_bytes), 0x00020000);

  // m -> (nb, md, mh, mw) by float reciprocals: (q + 0.5) / n is >= 0.5 / n away from an integer, more than the float error
  // for m < 2^21 (checked on the host).  Integer divisions here cost as much as the K loop of a one-tap tile.
  const float inv_mw = 1.0f / static_cast<float>(g.MW), inv_mh = 1.0f / static_cast<float>(g.MH), inv_md = 1.0f / static_cast<float>(g.MD);
  auto split_row = [&](int m, int& nb, int& md, int& mh, int& mw) {
    const int r1 = static_cast<int>((static_cast<float>(m) + 0.5f) * inv_mw);
    mw = m - r1 * g.MW;
    const int r2 = static_cast<int>((static_cast<float>(r1) + 0.5f) * inv_mh);
    mh = r1 - r2 * g.MH;
    nb = static_cast<int>((static_cast<float>(r2) + 0.5f) * inv_md);
    md = r2 - nb * g.MD;
  };
  // staging: thread -> (row, 16-byte piece): 4 rows of the A tile, NB of the B tile
  const int s_piece = tid & 7, s_row = tid >> 3;                  // rows s_row + 32 i
  unsigned a_base[4], b_base[NB];
  unsigned a_inv[4];               // bit e: the row's source of tap-list entry e lies in the padding (or the row does not exist);
                                   // bits >= ntaps are set, so that a piece past the end of K is "padding" too
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + s_row + 32 * i;
    unsigned ok = 0;
    a_base[i] = 0;
    if (m < M) {
      int nb, md, mh, mw;
      split_row(m, nb, md, mh, mw);
      const int sd = md * g.src_stride, sh = mh * g.src_stride, sw = mw * g.src_stride;
      a_base[i] = static_cast<unsigned>(((nb * g.SD + sd) * g.SH + sh) * g.SW + sw) * static_cast<unsigned>(g.Cin) * 2u;
      // entry index = (ed * nh + eh) * nw + ew: the validity map is the outer product of the per-axis ones
      unsigned wv = 0, hw = 0;
#pragma unroll
      for (int e = 0; e < 3; ++e)
        wv |= (e < nw && static_cast<unsigned>(sw + taps_delta(tp_w, e)) < static_cast<unsigned>(g.SW)) ? (1u << e) : 0u;
#pragma unroll
      for (int e = 0; e < 3; ++e)
        hw |= (e < nh && static_cast<unsigned>(sh + taps_delta(tp_h, e)) < static_cast<unsigned>(g.SH)) ? (wv << (e * nw)) : 0u;
#pragma unroll
      for (int e = 0; e < 3; ++e)
        ok |= (e < nd && static_cast<unsigned>(sd + taps_delta(tp_d, e)) < static_cast<unsigned>(g.SD)) ? (hw << (e * nh * nw)) : 0u;
    }
    a_inv[i] = ~ok;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int n = n0 + s_row + 32 * i;
    b_base[i] = n < g.Cout ? static_cast<unsigned>(n) * static_cast<unsigned>(g.Cin) * 2u : static_cast<unsigned>(kInvalid);
  }
  __syncthreads();                                                // tap tables

  const int PT = g.Cin >> 3;                                      // 16-byte pieces per tap
  const int pieces = ntaps * PT;
  const int KT = (pieces + 7) >> 3;
  const int kt_per = (KT + split - 1) / split;
  const int kt0 = ks * kt_per, kt1 = min(KT, kt0 + kt_per);
  const float inv_pt = 1.0f / static_cast<float>(PT);

  // Branch-free addressing (round 3; the first version tested three mask bits per row behind a branch each: 9.4 VALU
  // instructions per MFMA): the tap of this thread's piece indexes ONE packed table entry and ONE bit of the row's map;
  // an invalid A source gets bit 31 set (past the buffer: zeros), an invalid filter piece the offset kInvalid (2^30,
  // past any filter pack -- checked on the host -- also when row and piece are both invalid: 2^31).
  // Round 5: the operand tiles go global -> LDS by DMA.  Round 3 staged them through registers: 8 ds_write_b128 per thread
  // and K step, 13 cycles each on the CU's store path (profiles/r05_ubench_instruction_rates.txt) -- 36 % of the kernel's
  // time at two workgroups per CU, and 32 staging registers.  Lane L of wave w fills bytes [16 L, 16 L + 16) of the 1 KiB
  // that holds rows w * 8 + (L >> 3) + 32 i, i.e. the PHYSICAL piece s_piece of row s_row + 32 i; the swizzle key
  // ((row >> 1) & 7) is the same for a thread's rows (32 i >> 1 = 16 i), so the thread fetches ONE logical piece of the
  // flattened (tap, channel) axis per K step, as before.
  const int l_piece = s_piece ^ ((s_row >> 1) & 7);
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>((conv_lds_void*)&lds[0][0]));
  constexpr unsigned kStage = kTile + BNT * BK * 2;
  const unsigned dst_a = __builtin_amdgcn_readfirstlane(lds0 + static_cast<unsigned>(wave) * 1024u);
  auto dma_tile = [&](int kt, int stage) {
    const int j = kt * 8 + l_piece;                               // this thread's piece of the flattened (tap, channel) axis
    const int tap = min(static_cast<int>((static_cast<float>(j) + 0.5f) * inv_pt), 31);      // j >= pieces -> tap >= ntaps
    const int c8 = j - __mul24(tap, PT);
    const int2 tt = tap_tab[tap];
    const unsigned src = static_cast<unsigned>(tt.x + c8 * 16), wof = static_cast<unsigned>(tt.y + c8 * 16);
    const unsigned base = dst_a + static_cast<unsigned>(stage) * kStage;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      dma16(rx, (a_base[i] + src) | (((a_inv[i] >> tap) & 1u) << 31), base + i * 4096u);
#pragma unroll
    for (int i = 0; i < NB; ++i) dma16(rw, b_base[i] + wof, base + kTile + i * 4096u);
  };

  f32x16 acc[TN][2];          // [n tile][m tile] of the wave's part, D = [n][m]
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  const int fr = lane & 31, kg = lane >> 5;
  int fa_off[4][2], fb_off[4][TN];
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) {
    const int piece = 2 * k4 + kg;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rm = wm * 64 + i * 32 + fr;
      fa_off[k4][i] = rm * 128 + ((piece ^ ((rm >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int rn = wn * 32 * TN + i * 32 + fr;
      fb_off[k4][i] = rn * 128 + ((piece ^ ((rn >> 1) & 7)) << 4);
    }
  }
  auto compute = [&](int stage) {
    const unsigned char* ta = lds[stage];
    const unsigned char* tb = lds[stage] + kTile;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      s16x8 fa[2], fb[TN];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const s16x8*>(ta + fa_off[k4][i]);
#pragma unroll
      for (int i = 0; i < TN; ++i) fb[i] = *reinterpret_cast<const s16x8*>(tb + fb_off[k4][i]);
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mfma(fb[a], fa[b], acc[a][b]);      // D[n][m] += W[n][k] X[m][k]
    }
  };
  // Stage s is multiplied while the DMA of the next K step fills stage s ^ 1; the barrier at the end of a step says both
  // "everyone has read stage s" and (behind each wave's own dma_wait) "stage s ^ 1 has landed".
  if (kt0 < kt1) {
    dma_tile(kt0, 0);
    dma_wait();
    block_barrier();
    int kt = kt0;
    for (; kt + 2 <= kt1; kt += 2) {
      dma_tile(kt + 1, 1);
      compute(0);
      dma_wait();
      block_barrier();
      if (kt + 2 < kt1) dma_tile(kt + 2, 0);
      compute(1);
      dma_wait();
      block_barrier();
    }
    if (kt < kt1) {                                             // odd count: the last tile sits in stage 0
      compute(0);
      block_barrier();                                          // the epilogue reuses the stages
    }
  }

  // ---- epilogue
  const bool direct = g.out_stride == 1 && g.OD == g.MD && g.OH == g.MH && g.OW == g.MW;     // output row = m
  auto out_row = [&](int m) -> long {
    if (direct) return m;
    int nb, md, mh, mw;
    split_row(m, nb, md, mh, mw);
    return ((static_cast<long>(nb) * g.OD + md * g.out_stride + g.opd) * g.OH + mh * g.out_stride + g.oph) * g.OW + mw * g.out_stride + g.opw;
  };
  if (y32 != nullptr) {
    // fp32 partial tile of this split: a lane holds 4 consecutive n of its row -> 16-byte stores
    const long out_rows = static_cast<long>(g.N) * g.OD * g.OH * g.OW;
    float* part = y32 + static_cast<long>(ks) * out_rows * g.Cout;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int m = m0 + wm * 64 + b * 32 + fr;
      if (m >= M) continue;
      float* dst = part + out_row(m) * g.Cout;
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 32 * TN + a * 32 + 8 * q + 4 * kg;
          if (n < g.Cout)
            *reinterpret_cast<float4*>(dst + n) = float4{acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
        }
    }
    return;
  }
  // bf16: the wave's 64 x (32 TN) tile through its own LDS region, rows leave as whole 16-byte pieces (gemm.hip)
  constexpr int PITCH = 32 * TN * 2 + 16;
  unsigned char* stage = &lds[0][0] + wave * (64 * PITCH);
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = a * 32 + 8 * q + 4 * kg;
        const int n = n0 + wn * 32 * TN + nl;
        float4 bv{0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr && n < g.Cout) bv = *reinterpret_cast<const float4*>(bias + n);
        const float bq4[4] = {bv.x, bv.y, bv.z, bv.w};
        unsigned short h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = f32_to_bf16(acc[a][b][4 * q + e] + bq4[e]);
        *reinterpret_cast<uint2*>(stage + (b * 32 + fr) * PITCH + nl * 2) =
            uint2{static_cast<unsigned>(h[0]) | (static_cast<unsigned>(h[1]) << 16), static_cast<unsigned>(h[2]) | (static_cast<unsigned>(h[3]) << 16)};
      }
  constexpr int PP = 4 * TN, RP = 64 / PP;                       // 16-byte pieces per row of the wave tile, rows per pass
  const int piece = lane % PP, r0 = lane / PP;
#pragma unroll
  for (int it = 0; it < PP; ++it) {
    const int row = it * RP + r0;
    const int m = m0 + wm * 64 + row, n = n0 + wn * 32 * TN + piece * 8;
    if (m < M && n < g.Cout) {                                     // Cout is a multiple of 8 (checked on the host)
      const u32x4 v = *reinterpret_cast<const u32x4*>(stage + row * PITCH + piece * 16);
      *reinterpret_cast<u32x4*>(Y + out_row(m) * g.Cout + n) = v;
    }
  }
}

// y = bf16(sum over the `split` partial maps of y32 + bias): the last pass of a split-K convolution (rows x cout, cout % 8 == 0)
__global__ __launch_bounds__(256) void conv3d_finish_kernel(const float* __restrict__ y32, const float* __restrict__ bias,
                                                            unsigned short* __restrict__ y, long n8, int cout, int split) {
  const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n8) return;
  const int c = static_cast<int>((i * 8) % cout);
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < split; ++k) {
    const float4 a = reinterpret_cast<const float4*>(y32)[2 * (k * n8 + i)], b = reinterpret_cast<const float4*>(y32)[2 * (k * n8 + i) + 1];
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
  }
  if (bias != nullptr) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += bias[c + e];
  }
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = static_cast<unsigned>(f32_to_bf16(v[2 * e])) | (static_cast<unsigned>(f32_to_bf16(v[2 * e + 1])) << 16);
  reinterpret_cast<u32x4*>(y)[i] = o;
}

// ---------------------------------------------------------------------------------------------------------------
// part[chunk][slab(tap)][co][ci] = sum over the voxel rows m of the chunk:  dY[m][co] * X[src(m, tap)][ci]
// Block tile: (64 TW) co x (64 TW) ci for ONE tap; K = voxels in steps of 32 rows: both tiles are staged K-major
// ([row][channel], 128 TW-byte rows + 16 bytes of padding) and the MFMA fragments are cut out of them with the
// transposing ds_read_b64_tr_b16.  grid = voxel chunks x ci tiles x co tiles x taps.  Every block stores its fp32 tile
// with plain stores into its own chunk's partial map (no atomics); conv3d_wgrad_reduce sums the chunks.
// ---------------------------------------------------------------------------------------------------------------
constexpr int WK = 64;                      // voxel rows per K step
constexpr int kSegRows = 2048;              // voxel rows whose source offsets are tabulated at a time

template <int TW>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_kernel(
    const unsigned short* __restrict__ DY, const unsigned short* __restrict__ X, float* __restrict__ part, ConvGeom g,
    unsigned tp_d, unsigned tp_h, unsigned tp_w, int chunks, int rows_per_chunk, int tiles_co, int tiles_ci, int slabs,
    unsigned dy_bytes, unsigned x_bytes, float* __restrict__ bias_part) {
  constexpr int BT = 64 * TW;                // block tile side (channels)
  constexpr int WP = 2 * BT;                 // bytes per staged row: BT channels, no padding -- the 64-byte windows the transposing reads
                                             // take out of 4 consecutive rows are XOR-swizzled over the banks instead (row_swz)
  constexpr int PR = 8 * TW;                 // 16-byte pieces per row
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2 * WK * WP];       // [stage]: dY tile, then X tile
  __shared__ unsigned xrow[kSegRows];                                              // source row offsets of a segment of voxel rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nd = taps_count(tp_d), nh = taps_count(tp_h), nw = taps_count(tp_w);
  const int ntaps = nd * nh * nw;
  // block -> (ci tile, co tile, tap, chunk), chunk SLOWEST and every XCD walking a contiguous range: the workgroups that
  // run at the same time on an XCD are the tiles and taps of the same few voxel chunks, whose dY and x rows they share
  // through that XCD's L2.  (With the chunk fastest, co-resident workgroups shared nothing and the projection shapes ran
  // at the HBM rate of their re-reads: 2.9 GB per call for 234 000 x 384 x 1024.)
  long bid = xcd_contiguous(blockIdx.x, static_cast<long>(chunks) * tiles_ci * tiles_co * ntaps);
  if (bid < 0) return;
  const int tci = static_cast<int>(bid % tiles_ci); bid /= tiles_ci;
  const int tco = static_cast<int>(bid % tiles_co); bid /= tiles_co;
  const int tap = static_cast<int>(bid % ntaps); bid /= ntaps;
  const int chunk = static_cast<int>(bid);
  const int ed = tap / (nh * nw), r_ = tap - ed * nh * nw, eh = r_ / nw, ew = r_ - eh * nw;
  const int dd = taps_delta(tp_d, ed), dh = taps_delta(tp_h, eh), dw = taps_delta(tp_w, ew);
  const int slab = (taps_t(tp_d, ed) * 3 + taps_t(tp_h, eh)) * 3 + taps_t(tp_w, ew);
  const int M = g.N * g.MD * g.MH * g.MW;
  const int m_beg = chunk * rows_per_chunk, m_end = min(M, m_beg + rows_per_chunk);
  const int co0 = tco * BT, ci0 = tci * BT;
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(DY), 0, static_cast<int>(dy_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(X), 0, static_cast<int>(x_bytes), 0x00020000);

  // staging: a K step is WK rows x PR pieces per operand = 8 WK TW pieces of 16 bytes: NS per thread and operand
  constexpr int NS = WK * PR / 256, RS = 256 / PR;                // pieces per thread, rows per pass
  const int s_piece = tid % PR, s_row = tid / PR;                 // rows s_row + RS i
  // A transposing read takes a 64-byte window (32 channels) out of each of 4 consecutive rows: 4 x 16 banks.  With
  // 256-byte rows all four start in the same bank, with 128-byte rows every second one: the 16-byte pieces of a row are
  // stored XORed with 4 * (row & 3) (resp. 4 * ((row >> 1) & 1)), which turns the four windows into a tiling of the 64
  // banks.  (With 16 bytes of row padding instead, the windows overlapped 4-fold: 58 % of the LDS cycles were conflicts.)
  auto row_swz = [](int r) -> int { return TW == 2 ? (r & 3) << 2 : ((r >> 1) & 1) << 2; };
  // Round 5: both tiles go global -> LDS by DMA (see conv3d_igemm_kernel).  Lane L of wave w fills bytes [16 L, 16 L + 16) of
  // the 1 KiB that holds rows s_row + RS i (s_row = tid / PR: 64 / PR consecutive rows per wave), i.e. the PHYSICAL piece
  // s_piece of its row; the rows of a thread share their swizzle key (RS is a multiple of 4), so it fetches one logical
  // piece l_piece of the channel axis for all of them.
  const int l_piece = s_piece ^ row_swz(s_row);
  const bool co_ok = co0 + l_piece * 8 < g.Cout, ci_ok = ci0 + l_piece * 8 < g.Cin;
  const unsigned y_col = co_ok ? static_cast<unsigned>(co0 + l_piece * 8) * 2u : 0x80000000u;
  const unsigned x_col = ci_ok ? static_cast<unsigned>(ci0 + l_piece * 8) * 2u : 0x80000000u;
  // bias_part (a linear layer's weight gradient, Cin not a multiple of the tile): the column sums of dY -- the layer's bias
  // gradient -- come out of the same MFMAs as column Cin of the product, against a column of ONES standing in the X tile's
  // first padding piece.  That piece is written once here and left out of the DMA (its lanes are masked off); rows past
  // the chunk read dY = 0 and add nothing.  (A separate column-sum kernel re-read dY: 0.7 ms per Swin step.)
  const bool bias_here = bias_part != nullptr && tci == tiles_ci - 1;
  const bool ones_piece = bias_here && ci0 + l_piece * 8 == g.Cin;
  if (ones_piece) {
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int i = 0; i < WK * PR / 256; ++i)
        *reinterpret_cast<u32x4*>(&lds[st][WK * WP + (s_row + (256 / PR) * i) * WP + s_piece * 16]) = u32x4{0x00003f80u, 0u, 0u, 0u};
  }
  // The source row of every voxel row of a segment is worked out ONCE per workgroup into LDS (byte offset of the shifted x
  // voxel's first channel, or an out-of-range marker for the padding): with the decomposition m -> (nb, md, mh, mw) done
  // per K step and thread, the kernel issued 12.7 VALU instructions per MFMA and was VALU-bound (profiles/r03_conv_pmc.txt).
  const float inv_mw = 1.0f / static_cast<float>(g.MW), inv_mh = 1.0f / static_cast<float>(g.MH), inv_md = 1.0f / static_cast<float>(g.MD);
  auto fill_rows = [&](int seg_beg, int seg_end) {
    for (int m = seg_beg + tid; m < seg_end; m += 256) {
      // m -> (nb, md, mh, mw) by float reciprocals: (q + 0.5) / n is >= 0.5 / n away from an integer, more than the
      // float error for m < 2^21 (checked on the host)
      const int r1 = static_cast<int>((static_cast<float>(m) + 0.5f) * inv_mw), mw = m - r1 * g.MW;
      const int r2 = static_cast<int>((static_cast<float>(r1) + 0.5f) * inv_mh), mh = r1 - r2 * g.MH;
      const int nb = static_cast<int>((static_cast<float>(r2) + 0.5f) * inv_md), md = r2 - nb * g.MD;
      const int sd = md * g.src_stride + dd, sh = mh * g.src_stride + dh, sw = mw * g.src_stride + dw;
      const bool ok = static_cast<unsigned>(sd) < static_cast<unsigned>(g.SD) && static_cast<unsigned>(sh) < static_cast<unsigned>(g.SH) &&
                      static_cast<unsigned>(sw) < static_cast<unsigned>(g.SW);
      xrow[m - seg_beg] = ok ? static_cast<unsigned>(((nb * g.SD + sd) * g.SH + sh) * g.SW + sw) * static_cast<unsigned>(g.Cin) * 2u : 0x80000000u;
    }
  };
  int seg0 = 0, seg1 = 0;                                         // rows of the current segment
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>((conv_lds_void*)&lds[0][0]));
  const unsigned dst0 = __builtin_amdgcn_readfirstlane(lds0 + static_cast<unsigned>(wave) * 1024u);
  auto dma_step = [&](int m_base, int stage) {
    const unsigned base = dst0 + static_cast<unsigned>(stage) * (2u * WK * WP);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int m = m_base + s_row + RS * i;
      const bool in = m < seg1;
      const unsigned xo = in ? xrow[m - seg0] : 0x80000000u;
      // (a marker plus the column offset stays out of range: the buffers are < 2^31 bytes)
      dma16(rdy, in ? static_cast<unsigned>(m) * static_cast<unsigned>(g.Cout) * 2u + y_col : 0x80000000u, base + i * (RS * WP));
      if (!ones_piece) dma16(rx, xo + x_col, base + WK * WP + i * (RS * WP));
    }
  };
  f32x16 acc[TW][TW];          // [co tile][ci tile] of the wave's quadrant
#pragma unroll
  for (int a = 0; a < TW; ++a)
#pragma unroll
    for (int b = 0; b < TW; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  // transposing fragment read (msda3d_mma.hpp): within 16 lanes, lane supplies row (lane & 15) >> 2 of a 4-row set and
  // 4 channels; receives its channel's 4 rows.  Rows 8 kg + {0..3} and + 4 make the 8 K values of the lane's fragment.
  const int kg = lane >> 5;
  const int t_row = 8 * kg + ((lane & 15) >> 2), t_ch = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int t_swz = row_swz(t_row) * 16;         // rows t_row, t_row + 4, + 16 k16 share it
  auto frag = [&](const unsigned char* tile, int k16, int ch0) -> s16x8 {
    const unsigned char* p = tile + (k16 * 16 + t_row) * WP + (((ch0 + t_ch) * 2) ^ t_swz);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * WP));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto compute = [&](int stage) {
    const unsigned char* ty = lds[stage];
    const unsigned char* tx = lds[stage] + WK * WP;
#pragma unroll
    for (int k16 = 0; k16 < WK / 16; ++k16) {
      s16x8 fy[TW], fx[TW];
#pragma unroll
      for (int i = 0; i < TW; ++i) {
        fy[i] = frag(ty, k16, wm * 32 * TW + i * 32);
        fx[i] = frag(tx, k16, wn * 32 * TW + i * 32);
      }
#pragma unroll
      for (int a = 0; a < TW; ++a)
#pragma unroll
        for (int b = 0; b < TW; ++b) acc[a][b] = mfma(fy[a], fx[b], acc[a][b]);      // D[co][ci] += dY[k][co] X[k][ci]
    }
  };
  for (seg0 = m_beg; seg0 < m_end; seg0 += kSegRows) {
    seg1 = min(m_end, seg0 + kSegRows);
    __syncthreads();                                                // the previous segment's table and tiles are done with
    fill_rows(seg0, seg1);
    __syncthreads();
    const int steps = (seg1 - seg0 + WK - 1) / WK;
    dma_step(seg0, 0);
    dma_wait();
    block_barrier();
    for (int s = 0; s < steps; s += 2) {
      if (s + 1 < steps) dma_step(seg0 + (s + 1) * WK, 1);
      compute(0);
      dma_wait();
      block_barrier();
      if (s + 1 >= steps) break;
      if (s + 2 < steps) dma_step(seg0 + (s + 2) * WK, 0);
      compute(1);
      dma_wait();
      block_barrier();
    }
  }
  // D layout: register r of a lane = row (r & 3) + 8 (r >> 2) + 4 kg (co), column lane & 31 (ci): 128-byte rows
  float* dst = part + (static_cast<long>(chunk) * slabs + (slabs == 1 ? 0 : slab)) * g.Cout * g.Cin;      // slabs == 1: the one tap of a projection
#pragma unroll
  for (int a = 0; a < TW; ++a)
#pragma unroll
    for (int b = 0; b < TW; ++b) {
      const int ci = ci0 + wn * 32 * TW + b * 32 + (lane & 31);
      if (bias_here && ci == g.Cin) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + wm * 32 * TW + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
          if (co < g.Cout) bias_part[static_cast<long>(chunk) * g.Cout + co] = acc[a][b][r];
        }
      }
      if (ci >= g.Cin) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 32 * TW + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        if (co < g.Cout) dst[static_cast<long>(co) * g.Cin + ci] = acc[a][b][r];
      }
    }
}

// dw[co][ci][tap] (slabs == 27: nn.Conv3d's weight layout) or dw[co][ci] (slabs == 1) = sum over the chunks of
// part[chunk][slab][co][ci].  A block takes kReducePairs consecutive (co, ci) pairs x all slabs: the partial maps are read
// as float4 along ci (512-byte runs per slab and chunk, all loads of a thread independent), the results are turned
// through LDS and leave as one contiguous run of float4.  (First version: 32 pairs per block, scalar loads, scalar
// stores -- 0.89 ms per step for 29 launches; coci is a multiple of 64, both channel counts being multiples of 8.)
constexpr int kReducePairs = 128;
__global__ __launch_bounds__(256) void conv3d_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int chunks,
                                                                  int slabs, long coci) {
  __shared__ __attribute__((aligned(16))) float sh[kReducePairs * 27];
  const long e0 = static_cast<long>(blockIdx.x) * kReducePairs;
  const int quads = kReducePairs / 4;
  const int work = quads * slabs;                    // float4 columns of this block: 32 for a linear layer, 864 for 27 taps
  if (work < 256) {
    // a linear layer's gradient (slabs == 1): 32 columns -- with one thread per column, 7 of 8 threads idled while the other
    // walked up to 1024 chunks one dependent load after the other (24 us per launch, 57 launches per Swin step).  The
    // chunks are dealt to 256 / work thread groups instead; LDS adds their sums.
    const int P = 256 / work;
    const int sub = threadIdx.x / work, idx = threadIdx.x - sub * work;
    const int tap = idx / quads, q = idx - tap * quads;
    float4 v{0.f, 0.f, 0.f, 0.f};
    if (sub < P && e0 + 4 * q < coci) {
      const float* src = part + static_cast<long>(tap) * coci + e0 + 4 * q;
      const long step = static_cast<long>(slabs) * coci;
      int k = sub;
      for (; k + 3 * P < chunks; k += 4 * P) {
        const float4 a = *reinterpret_cast<const float4*>(src + (k + 0 * P) * step), b = *reinterpret_cast<const float4*>(src + (k + 1 * P) * step);
        const float4 c = *reinterpret_cast<const float4*>(src + (k + 2 * P) * step), d = *reinterpret_cast<const float4*>(src + (k + 3 * P) * step);
        v.x += (a.x + b.x) + (c.x + d.x); v.y += (a.y + b.y) + (c.y + d.y);
        v.z += (a.z + b.z) + (c.z + d.z); v.w += (a.w + b.w) + (c.w + d.w);
      }
      for (; k < chunks; k += P) {
        const float4 a = *reinterpret_cast<const float4*>(src + k * step);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
    }
    float4* turn = reinterpret_cast<float4*>(sh);                  // [group][column]: 256 float4 at most
    if (sub < P) turn[sub * work + idx] = v;
    __syncthreads();
    if (sub == 0) {
      for (int g2 = 1; g2 < P; ++g2) {
        const float4 o = turn[g2 * work + idx];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
    }
    __syncthreads();
    if (sub == 0) {
      sh[(4 * q + 0) * slabs + tap] = v.x;
      sh[(4 * q + 1) * slabs + tap] = v.y;
      sh[(4 * q + 2) * slabs + tap] = v.z;
      sh[(4 * q + 3) * slabs + tap] = v.w;
    }
  } else {
  for (int idx = threadIdx.x; idx < quads * slabs; idx += 256) {
    const int tap = idx / quads, q = idx - tap * quads;
    float4 v{0.f, 0.f, 0.f, 0.f};
    if (e0 + 4 * q < coci) {
      const float* src = part + static_cast<long>(tap) * coci + e0 + 4 * q;
      const long step = static_cast<long>(slabs) * coci;
      int k = 0;
      for (; k + 4 <= chunks; k += 4) {
        const float4 a = *reinterpret_cast<const float4*>(src + (k + 0) * step), b = *reinterpret_cast<const float4*>(src + (k + 1) * step);
        const float4 c = *reinterpret_cast<const float4*>(src + (k + 2) * step), d = *reinterpret_cast<const float4*>(src + (k + 3) * step);
        v.x += (a.x + b.x) + (c.x + d.x); v.y += (a.y + b.y) + (c.y + d.y);
        v.z += (a.z + b.z) + (c.z + d.z); v.w += (a.w + b.w) + (c.w + d.w);
      }
      for (; k < chunks; ++k) {
        const float4 a = *reinterpret_cast<const float4*>(src + k * step);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
    }
    sh[(4 * q + 0) * slabs + tap] = v.x;
    sh[(4 * q + 1) * slabs + tap] = v.y;
    sh[(4 * q + 2) * slabs + tap] = v.z;
    sh[(4 * q + 3) * slabs + tap] = v.w;
  }
  }
  __syncthreads();
  const long total = coci * slabs, base = e0 * slabs;             // both multiples of 4
  for (int idx = threadIdx.x; idx < quads * slabs; idx += 256)
    if (base + 4 * idx < total) *reinterpret_cast<float4*>(dw + base + 4 * idx) = *reinterpret_cast<const float4*>(&sh[4 * idx]);
}

// nn.Conv3d's fp32 weight (Cout, Cin, 27) -> the two bf16 filter packs of the kernels above in one pass: wk (27, Cout, Cin)
// for the forward and the weight gradient's layout, wkt (27, Cin, Cout) for the data gradient.  A block turns a 32 x 32
// (co, ci) tile through LDS so that both outputs leave as contiguous 64-byte runs.
__global__ __launch_bounds__(256) void conv3d_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ wk,
                                                          unsigned short* __restrict__ wkt, int co_n, int ci_n) {
  __shared__ unsigned short sh[27][32][33];
  const int tiles_ci = (ci_n + 31) / 32;
  const int co0 = (blockIdx.x / tiles_ci) * 32, ci0 = (blockIdx.x % tiles_ci) * 32;
  // 32 x 32 pairs x 27 taps = 27648 floats; the 27 taps of a pair are contiguous, pairs along ci are 27 floats apart
  for (int idx = threadIdx.x; idx < 32 * 32 * 27; idx += 256) {
    const int col = idx % (32 * 27), a = idx / (32 * 27);          // a: co inside the tile; col: (ci, tap) run of 864 floats
    const int b = col / 27, tap = col - b * 27;
    const int co = co0 + a, ci = ci0 + b;
    float v = 0.f;
    if (co < co_n && ci < ci_n) v = w[(static_cast<long>(co) * ci_n + ci) * 27 + tap];
    sh[tap][a][b] = f32_to_bf16(v);
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 27 * 32 * 32; idx += 256) {
    const int x = idx & 31, y = (idx >> 5) & 31, tap = idx >> 10;
    if (co0 + y < co_n && ci0 + x < ci_n) wk[(static_cast<long>(tap) * co_n + co0 + y) * ci_n + ci0 + x] = sh[tap][y][x];
    if (wkt != nullptr && ci0 + y < ci_n && co0 + x < co_n) wkt[(static_cast<long>(tap) * ci_n + ci0 + y) * co_n + co0 + x] = sh[tap][x][y];
  }
}

// The same for MANY layers in one launch (one block per 32 x 32 tile of any layer): the per-layer launches are 2 to 576
// blocks each -- the small layers do not fill the chip and the 14 layers of the flagship model cost 0.6 ms per step as
// 56 torch permute / cast copies.  table[l] = {w, wk, wkt, Cout | Cin << 32, first tile}; tiles are numbered layer by layer.
struct PackEntry {
  const float* w;
  unsigned short* wk;
  unsigned short* wkt;
  long co_ci;
  long tile_begin;
};
__global__ __launch_bounds__(256) void conv3d_pack_many_kernel(const PackEntry* __restrict__ table, int n_layers) {
  // second version: float4 reads of the (ci, tap) runs, 16-byte stores of both packs (the first one moved 2 bytes per
  // thread and store: 0.33 ms per step for 280 MB)
  constexpr int P = 34;                                           // row pitch in bf16: rows start 4-byte aligned
  __shared__ __attribute__((aligned(16))) unsigned short sh[27][32][P];
  int l = 0;
  while (l + 1 < n_layers && static_cast<long>(blockIdx.x) >= table[l + 1].tile_begin) ++l;      // block-uniform
  const PackEntry e = table[l];
  const int co_n = static_cast<int>(e.co_ci & 0xffffffffL), ci_n = static_cast<int>(e.co_ci >> 32);
  const int tile = static_cast<int>(blockIdx.x - e.tile_begin);
  const int tiles_ci = (ci_n + 31) / 32;
  const int co0 = (tile / tiles_ci) * 32, ci0 = (tile % tiles_ci) * 32;
  const int ci_valid = min(32, ci_n - ci0);                       // a multiple of 8: the valid run of a co row is a multiple of 4 floats
  // 32 co rows x 216 float4: a co row's (ci, tap) values are one contiguous run of ci_valid * 27 floats
  for (int idx = threadIdx.x; idx < 32 * 216; idx += 256) {
    const int a = idx / 216, f = idx - a * 216;
    float4 v{0.f, 0.f, 0.f, 0.f};
    if (co0 + a < co_n && 4 * f < ci_valid * 27)
      v = *reinterpret_cast<const float4*>(e.w + (static_cast<long>(co0 + a) * ci_n + ci0) * 27 + 4 * f);
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = 4 * f + j, b = p / 27, tap = p - b * 27;
      sh[tap][a][b] = f32_to_bf16(vv[j]);
    }
  }
  __syncthreads();
  // 27 taps x 32 rows x 4 pieces of 8 values, for both packs
  for (int idx = threadIdx.x; idx < 27 * 32 * 4; idx += 256) {
    const int x8 = idx & 3, y = (idx >> 2) & 31, tap = idx >> 7;
    if (co0 + y < co_n && ci0 + 8 * x8 < ci_n) {                  // wk[tap][co0 + y][ci0 + 8 x8 ..]
      const unsigned* src = reinterpret_cast<const unsigned*>(&sh[tap][y][8 * x8]);
      *reinterpret_cast<u32x4*>(e.wk + (static_cast<long>(tap) * co_n + co0 + y) * ci_n + ci0 + 8 * x8) = u32x4{src[0], src[1], src[2], src[3]};
    }
    if (e.wkt != nullptr && ci0 + y < ci_n && co0 + 8 * x8 < co_n) {      // wkt[tap][ci0 + y][co0 + 8 x8 ..] = sh[tap][8 x8 + j][y]
      unsigned r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        r[j] = static_cast<unsigned>(sh[tap][8 * x8 + 2 * j][y]) | (static_cast<unsigned>(sh[tap][8 * x8 + 2 * j + 1][y]) << 16);
      *reinterpret_cast<u32x4*>(e.wkt + (static_cast<long>(tap) * ci_n + ci0 + y) * co_n + co0 + 8 * x8) = u32x4{r[0], r[1], r[2], r[3]};
    }
  }
}

#include "conv_dgrad_s2.hpp"
#include "conv_wgrad_ring.hpp"

int check_geom(const ConvGeom& g) {
  if (g.N <= 0 || g.SD <= 0 || g.SH <= 0 || g.SW <= 0 || g.MD <= 0 || g.MH <= 0 || g.MW <= 0 || g.OD <= 0 || g.OH <= 0 || g.OW <= 0 ||
      g.Cin <= 0 || g.Cout <= 0 || (g.Cin & 7) || (g.Cout & 7) || g.src_stride < 1 || g.src_stride > 2 || g.out_stride < 1 || g.out_stride > 2)
    return TRANSOAR_CONVGEMM_ERR_DIM;
  const long src = static_cast<long>(g.N) * g.SD * g.SH * g.SW * g.Cin * 2, out = static_cast<long>(g.N) * g.OD * g.OH * g.OW * g.Cout * 2;
  const long rows = static_cast<long>(g.N) * g.MD * g.MH * g.MW;
  if (src >= 0x7ffffff0L || out >= 0x7ffffff0L * 2 || rows >= (1L << 31) || rows * g.Cout * 2 >= 0x7ffffff0L * 2 || 27L * g.Cout * g.Cin * 2 >= 0x3ffffff0L)
    return TRANSOAR_CONVGEMM_ERR_DIM;
  return 0;
}
bool taps_ok(unsigned p) { return (p & 3) >= 1 && (p & 3) <= 3; }

}  // namespace

extern "C" int transoar_conv3d_igemm(const void* x, const void* wk, const float* bias, void* y, float* y32, int N, int SD,
                                     int SH, int SW, int Cin, int Cout, int MD, int MH, int MW, int src_stride, int OD, int OH,
                                     int OW, int out_stride, int opd, int oph, int opw, unsigned taps_d, unsigned taps_h,
                                     unsigned taps_w, int split, int classes, void* hip_stream) {
  if (!x || !wk || (!y && !y32)) return TRANSOAR_CONVGEMM_ERR_NULL;
  if (classes) {               // all parity classes of a stride-2 data gradient: the row space is derived per class
    MD = (OD + 1) / 2; MH = (OH + 1) / 2; MW = (OW + 1) / 2;
    src_stride = 1; out_stride = 2; opd = oph = opw = 0;
    taps_d = taps_h = taps_w = kTapsParity1;
  }
  const ConvGeom g{N, SD, SH, SW, MD, MH, MW, OD, OH, OW, src_stride, out_stride, opd, oph, opw, Cin, Cout};
  const int rc = check_geom(g);
  if (rc) return rc;
  if (!taps_ok(taps_d) || !taps_ok(taps_h) || !taps_ok(taps_w) || split < 1 || (split > 1 && !y32)) return TRANSOAR_CONVGEMM_ERR_DIM;
  if (static_cast<long>(N) * MD * MH * MW >= (1L << 21)) return TRANSOAR_CONVGEMM_ERR_DIM;       // row decomposition by float reciprocals
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const bool narrow = Cout <= 64;                       // 128 x 64 tiles: up to 64 output channels do not half-fill a 128-wide one
  const int bn = narrow ? 64 : 128;
  const int tiles_n = (Cout + bn - 1) / bn;
  long n_tiles = 0;
  if (classes) {
    for (int c = 0; c < 8; ++c) {
      const long md = (OD - ((c >> 2) & 1) + 1) / 2, mh = (OH - ((c >> 1) & 1) + 1) / 2, mw = (OW - (c & 1) + 1) / 2;
      n_tiles += (N * md * mh * mw + BM - 1) / BM * tiles_n;
    }
  } else {
    n_tiles = (static_cast<long>(N) * MD * MH * MW + BM - 1) / BM * tiles_n;
  }
  const long blocks = ((n_tiles * split + 7) / 8) * 8;
  const unsigned xb = static_cast<unsigned>(static_cast<long>(N) * SD * SH * SW * Cin * 2), wb = static_cast<unsigned>(27L * Cout * Cin * 2);
  if (narrow)
    hipLaunchKernelGGL(conv3d_igemm_kernel<1>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, static_cast<const unsigned short*>(x),
                       static_cast<const unsigned short*>(wk), bias, static_cast<unsigned short*>(y), y32, g, taps_d, taps_h, taps_w,
                       split, n_tiles, tiles_n, classes, xb, wb);
  else
    hipLaunchKernelGGL(conv3d_igemm_kernel<2>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, static_cast<const unsigned short*>(x),
                       static_cast<const unsigned short*>(wk), bias, static_cast<unsigned short*>(y), y32, g, taps_d, taps_h, taps_w,
                       split, n_tiles, tiles_n, classes, xb, wb);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_conv3d_dgrad_s2_halo(const void* dy, const void* wkt, void* dx, int N, int OD, int OH, int OW, int D, int H, int W,
                                            int Cin, int Cout, void* hip_stream) {
  if (!dy || !wkt || !dx) return TRANSOAR_CONVGEMM_ERR_NULL;
  if (N <= 0 || OD <= 0 || OH <= 0 || OW <= 0 || Cin <= 0 || Cin > 32 || (Cin & 7) || (Cout != 16 && Cout != 32 && Cout != 48))
    return TRANSOAR_CONVGEMM_ERR_DIM;
  if ((D != 2 * OD && D != 2 * OD - 1) || (H != 2 * OH && H != 2 * OH - 1) || (W != 2 * OW && W != 2 * OW - 1)) return TRANSOAR_CONVGEMM_ERR_DIM;
  const long dy_bytes = static_cast<long>(N) * OD * OH * OW * Cout * 2;
  if (dy_bytes >= 0x7ffffff0L || (4L * H + 8) * W * Cin * 2 >= 0x7ffffff0L) return TRANSOAR_CONVGEMM_ERR_DIM;     // 32-bit offsets inside a tile
  const int td = (OD + kDs2TD - 1) / kDs2TD, th = (OH + kDs2TH - 1) / kDs2TH, tw = (OW + kDs2TW - 1) / kDs2TW;
  const long n_tiles = static_cast<long>(N) * td * th * tw;
  // persistent: exactly one resident set of workgroups (a partial second set would run alone at the end)
  static int resident[3] = {0, 0, 0};
  const int ki = Cout / 16 - 1;
  if (!resident[ki]) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    const void* fn = ki == 0 ? reinterpret_cast<const void*>(conv3d_dgrad_s2_halo_kernel<1>)
                   : ki == 1 ? reinterpret_cast<const void*>(conv3d_dgrad_s2_halo_kernel<2>) : reinterpret_cast<const void*>(conv3d_dgrad_s2_halo_kernel<3>);
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1)
      return TRANSOAR_CONVGEMM_ERR_DIM;
    resident[ki] = per_cu * prop.multiProcessorCount;
  }
  const unsigned blocks = static_cast<unsigned>(n_tiles < resident[ki] ? n_tiles : resident[ki]);
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
#define TRANSOAR_DS2_LAUNCH(KS)                                                                                                     \
  hipLaunchKernelGGL(conv3d_dgrad_s2_halo_kernel<KS>, dim3(blocks), dim3(256), 0, st, static_cast<const unsigned short*>(dy),       \
                     static_cast<const unsigned short*>(wkt), static_cast<unsigned short*>(dx), N, OD, OH, OW, D, H, W, Cin, td, th, tw, \
                     static_cast<unsigned>(dy_bytes))
  if (Cout == 16) TRANSOAR_DS2_LAUNCH(1);
  else if (Cout == 32) TRANSOAR_DS2_LAUNCH(2);
  else TRANSOAR_DS2_LAUNCH(3);
#undef TRANSOAR_DS2_LAUNCH
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_conv3d_finish(const float* y32, const float* bias, void* y, long rows, int cout, int split, void* hip_stream) {
  if (!y32 || !y) return TRANSOAR_CONVGEMM_ERR_NULL;
  if (rows <= 0 || cout <= 0 || (cout & 7) || split < 1) return TRANSOAR_CONVGEMM_ERR_DIM;
  const long n8 = rows * cout / 8;
  hipLaunchKernelGGL(conv3d_finish_kernel, dim3(static_cast<unsigned>((n8 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                     y32, bias, static_cast<unsigned short*>(y), n8, cout, split);
  return static_cast<int>(hipGetLastError());
}

extern "C" long transoar_conv3d_wgrad_part_floats(int Cin, int Cout, int chunks, int taps_out) { return static_cast<long>(taps_out) * Cin * Cout * chunks; }

static int wgrad_launch(const void* dy, const void* x, float* part, float* dw, int N, int SD, int SH, int SW, int Cin,
                        int Cout, int MD, int MH, int MW, int src_stride, unsigned taps_d, unsigned taps_h,
                        unsigned taps_w, int chunks, int taps_out, float* bias_part, float* db, void* hip_stream) {
  if (!dy || !x || !part || !dw) return TRANSOAR_CONVGEMM_ERR_NULL;
  const ConvGeom g{N, SD, SH, SW, MD, MH, MW, MD, MH, MW, src_stride, 1, 0, 0, 0, Cin, Cout};
  const int rc = check_geom(g);
  if (rc) return rc;
  if (!taps_ok(taps_d) || !taps_ok(taps_h) || !taps_ok(taps_w) || chunks < 1 || (taps_out != 1 && taps_out != 27)) return TRANSOAR_CONVGEMM_ERR_DIM;
  const long M = static_cast<long>(N) * MD * MH * MW;
  if (M >= (1L << 21)) return TRANSOAR_CONVGEMM_ERR_DIM;          // row decomposition by float reciprocals (conv3d_wgrad_kernel)
  const int max_chunks = chunks;
  if (chunks > M) chunks = static_cast<int>(M);
  const int rows_per_chunk = static_cast<int>(((M + chunks - 1) / chunks + WK - 1) / WK * WK);
  chunks = static_cast<int>((M + rows_per_chunk - 1) / rows_per_chunk);
  if (chunks > max_chunks) return TRANSOAR_CONVGEMM_ERR_DIM;
  const int ntaps = static_cast<int>((taps_d & 3) * (taps_h & 3) * (taps_w & 3));
  if (taps_out == 1 && ntaps != 1) return TRANSOAR_CONVGEMM_ERR_DIM;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const bool small = Cin <= 64 && Cout <= 64;              // 64 x 64 tiles for the 24..64-channel layers
  const int bt = small ? 64 : 128;
  const int tiles_co = (Cout + bt - 1) / bt, tiles_ci = (Cin + bt - 1) / bt;
  const long blocks = (static_cast<long>(chunks) * tiles_ci * tiles_co * ntaps + 7) / 8 * 8;
  if (blocks >= (1L << 31)) return TRANSOAR_CONVGEMM_ERR_DIM;
  const unsigned dyb = static_cast<unsigned>(M * Cout * 2), xb = static_cast<unsigned>(static_cast<long>(N) * SD * SH * SW * Cin * 2);
  if (small)
    hipLaunchKernelGGL(conv3d_wgrad_kernel<1>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, static_cast<const unsigned short*>(dy),
                       static_cast<const unsigned short*>(x), part, g, taps_d, taps_h, taps_w, chunks, rows_per_chunk, tiles_co, tiles_ci, taps_out, dyb, xb, bias_part);
  else
    hipLaunchKernelGGL(conv3d_wgrad_kernel<2>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, static_cast<const unsigned short*>(dy),
                       static_cast<const unsigned short*>(x), part, g, taps_d, taps_h, taps_w, chunks, rows_per_chunk, tiles_co, tiles_ci, taps_out, dyb, xb, bias_part);
  const long coci = static_cast<long>(Cout) * Cin;
  hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel, dim3(static_cast<unsigned>((coci + kReducePairs - 1) / kReducePairs)), dim3(256), 0, st, part, dw, chunks, taps_out, coci);
  if (bias_part != nullptr)          // the same reduction over the chunks for the (chunks, Cout) bias partials
    hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel, dim3(static_cast<unsigned>((Cout + kReducePairs - 1) / kReducePairs)), dim3(256), 0, st, bias_part, db, chunks, 1,
                       static_cast<long>(Cout));
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_conv3d_wgrad(const void* dy, const void* x, float* part, float* dw, int N, int SD, int SH, int SW, int Cin,
                                     int Cout, int MD, int MH, int MW, int src_stride, unsigned taps_d, unsigned taps_h,
                                     unsigned taps_w, int chunks, int taps_out, void* hip_stream) {
  return wgrad_launch(dy, x, part, dw, N, SD, SH, SW, Cin, Cout, MD, MH, MW, src_stride, taps_d, taps_h, taps_w, chunks, taps_out, nullptr, nullptr,
                      hip_stream);
}

extern "C" int transoar_linear_wgrad_bias(const void* dy, const void* x, float* part, float* dw, float* bias_part, float* db, int T,
                                          int Cin, int Cout, int chunks, void* hip_stream) {
  if (!bias_part || !db) return TRANSOAR_CONVGEMM_ERR_NULL;
  const int bt = (Cin <= 64 && Cout <= 64) ? 64 : 128;
  if (Cin % bt == 0) return TRANSOAR_CONVGEMM_ERR_DIM;            // no padding column in the last tile to carry the ones
  constexpr unsigned kOneTap = 1u | (1u << 2) | (1u << 4);          // one tap: delta 0, slot 1 (conv_gemm.py TAPS_ONE)
  return wgrad_launch(dy, x, part, dw, 1, 1, 1, T, Cin, Cout, 1, 1, T, 1, kOneTap, kOneTap, kOneTap, chunks, 1, bias_part, db, hip_stream);
}

extern "C" int transoar_conv3d_wgrad_ring(const void* dy, const void* x, float* part, float* dw, int N, int SD, int SH, int SW, int Cin,
                                          int Cout, int MD, int MH, int MW, int src_stride, int chunks, void* hip_stream) {
  if (!dy || !x || !part || !dw) return TRANSOAR_CONVGEMM_ERR_NULL;
  if (N <= 0 || SD <= 0 || SH <= 0 || SW <= 0 || MD <= 0 || MH <= 0 || MW <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 7) || (Cout & 7) ||
      Cin > 64 || Cout > 64 || (MW & 63) || chunks < 1 || (src_stride != 1 && src_stride != 2))
    return TRANSOAR_CONVGEMM_ERR_DIM;
  if (MD != (SD - 1) / src_stride + 1 || MH != (SH - 1) / src_stride + 1 || MW != (SW - 1) / src_stride + 1) return TRANSOAR_CONVGEMM_ERR_DIM;
  const int tiles_co = (Cout + 31) / 32, tiles_ci = (Cin + 31) / 32;
  if (tiles_co * tiles_ci < 2) return TRANSOAR_CONVGEMM_ERR_DIM;        // 8 waves = 2 or 4 channel-tile pairs x a K split (<= 32 x 32: conv3d.hip's kernel)
  // tasks = (h chunk, W segment, (b, jd) slice): at least 4 per workgroup of a plane, chunks of >= 8 rows
  const long columns = static_cast<long>(N) * MD * (MW / 64);
  int h_chunks = 1;
  while (columns * h_chunks < 4L * chunks && (MH + h_chunks) / (h_chunks + 1) >= 8) ++h_chunks;
  const int h_chunk = (MH + h_chunks - 1) / h_chunks;
  h_chunks = (MH + h_chunk - 1) / h_chunk;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const dim3 grid(static_cast<unsigned>(chunks * 3));
  auto dys = static_cast<const unsigned short*>(dy);
  auto xs = static_cast<const unsigned short*>(x);
#define TRANSOAR_RING_LAUNCH(S_, CIT_)                                                                                                    \
  hipLaunchKernelGGL((conv3d_wgrad_ring_kernel<S_, CIT_>), grid, dim3(512), 0, st, dys, xs, part, N, SD, SH, SW, MD, MH, MW, Cin, Cout, tiles_co, \
                     tiles_ci, h_chunks, h_chunk)
  if (src_stride == 1) { if (tiles_ci == 1) TRANSOAR_RING_LAUNCH(1, 1); else TRANSOAR_RING_LAUNCH(1, 2); }
  else { if (tiles_ci == 1) TRANSOAR_RING_LAUNCH(2, 1); else TRANSOAR_RING_LAUNCH(2, 2); }
#undef TRANSOAR_RING_LAUNCH
  const long coci = static_cast<long>(Cout) * Cin;
  hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel, dim3(static_cast<unsigned>((coci + kReducePairs - 1) / kReducePairs)), dim3(256), 0, st, part, dw, chunks, 27, coci);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_conv3d_pack(const float* w, void* wk, void* wkt, int Cout, int Cin, void* hip_stream) {
  if (!w || !wk) return TRANSOAR_CONVGEMM_ERR_NULL;
  if (Cout <= 0 || Cin <= 0) return TRANSOAR_CONVGEMM_ERR_DIM;
  const int blocks = ((Cout + 31) / 32) * ((Cin + 31) / 32);
  hipLaunchKernelGGL(conv3d_pack_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), w,
                     static_cast<unsigned short*>(wk), static_cast<unsigned short*>(wkt), Cout, Cin);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_conv3d_pack_many(const void* table, int n_layers, long total_tiles, void* hip_stream) {
  if (!table) return TRANSOAR_CONVGEMM_ERR_NULL;
  if (n_layers <= 0 || total_tiles <= 0 || total_tiles >= (1L << 31)) return TRANSOAR_CONVGEMM_ERR_DIM;
  hipLaunchKernelGGL(conv3d_pack_many_kernel, dim3(static_cast<unsigned>(total_tiles)), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                     static_cast<const PackEntry*>(table), n_layers);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_convgemm_abi_version(void) { return 3; }
