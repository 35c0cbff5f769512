// Fused masked cross-attention of the Focused Decoder over per-organ gathered tokens (gfx950).
//
// Reference semantics: FocusedAttn.forward, transoar/models/necks/focused_decoder.py:228-262 (scores over all keys,
// -inf outside the organ's attn_area :138-159,243-247, softmax, weighted sum of the values).  The host side
// (transoar_amd/focused_decoder.py) has already gathered each organ's RoI tokens and folded the K / V projections into
// the queries, so what is left per group g = (batch, organ) is ONE plain attention
//
//     ctx[g] = softmax( mask( qf[g] (R x C)  .  k[g]^T (C x L) ) )  .  v[g] (L x C)          R = 216, C = 384, L <= 5520
//
// whose backward is  dqf = dS k,  dtok = dS^T qf + P^T dctx  with  dS = P o (dctx v^T - rowsum(dctx o ctx)).
// Round 3 ran it as torch bmm / masked_fill_ / softmax / baddbmm_ (hipBLASLt + aten, a bf16 score tensor in HBM);
// here QK^T -> mask -> softmax -> PV is one kernel (flash style: online softmax, no score tensor), and the backward
// recomputes P from the saved log-sum-exp in two kernels: a query-stationary one for dqf and a key-stationary one for
// dtok (each output is owned by one workgroup: no atomics).
//
// Common structure of the three kernels, shaped by C = 384 (a 32-row operand tile is 24 KB, an accumulator tile
// 32 x 384 fp32 is 192 registers):
//   * 256 threads = 4 waves, ONE wave per SIMD with the whole 512-register file: the stationary operands of a wave's
//     32 rows (24 MFMA B fragments each: 96 registers) and its 32 x 384 accumulator live in registers; only the streamed
//     operand tiles go through LDS;
//   * streamed tiles (32 rows x 768 B) arrive by LDS-DMA (global_load_lds, 16 B per lane, 1 KiB per wave instruction,
//     no staging registers) into a 2-deep ring, one barrier per tile;
//   * one LDS image serves both MFMA fragment reads: 16-byte piece p of row n sits at piece p ^ (4 (n & 3) | (n >> 2) & 3)
//     -- the 16 rows of a ds_read_b128 lane group land in 16 different 4-bank slots, and the four 64-byte windows of a
//     transposing ds_read_b64_tr_b16 (rows n .. n + 3) tile the 64 banks.  The image is lane-linear for the DMA; the
//     permutation is applied to the per-lane SOURCE address;
//   * scores are computed transposed (S^T = K Q^T) so that a lane holds 16 keys of ONE query row: row maxima / sums
//     are in-lane plus one exchange between the wave halves, and P^T goes back into the matrix cores as a B operand
//     after v_cvt_pk_bf16_f32 + v_permlane32_swap (no LDS round trip).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "../../include/transoar_attn.h"

#include "mfma_stream.hpp"

namespace transoar {

// key range of this workgroup: tiles [t0, t1) of the organ's n_tiles, split into n_split equal chunks
__device__ __forceinline__ void split_range(int tiles, int n_split, int split, int& t0, int& t1) {
  const int chunk = (tiles + n_split - 1) / n_split;
  t0 = min(split * chunk, tiles);
  t1 = min(t0 + chunk, tiles);
}

constexpr float kDefer = 8.f;               // log2 units: see roi_attn_fwd

#define TRANSOAR_ATTN_KERNEL __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))

// ---------------------------------------------------------------------------
// forward: grid (n_split, row blocks of 128, G)
//   n_split == 1: ctx (bf16) and lse written directly; else unnormalised partials (pacc fp32, pm, pl) for attn_fwd_combine
// ---------------------------------------------------------------------------
TRANSOAR_ATTN_KERNEL void roi_attn_fwd(
    const unsigned short* __restrict__ q, const unsigned short* __restrict__ k, const unsigned short* __restrict__ v,
    const unsigned* __restrict__ keybits, const int* __restrict__ n_tiles, unsigned short* __restrict__ ctx,
    float* __restrict__ lse, float* __restrict__ pacc, float* __restrict__ pm, float* __restrict__ pl,
    int O, int R, int L, long total_keys, int bits_stride, int n_split) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * kTile];
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const int split = blockIdx.x, rb = blockIdx.y, g = blockIdx.z;
  const int o = g % O;
  const int kh = lane >> 5;
  FragBase fs = frag_base(lane);          // fragment bases of the ring stage being read (stage 0 first)
  const int row = rb * 128 + wave * 32 + (lane & 31);
  const bool row_ok = row < R;
  int t0, t1;
  split_range(n_tiles[o], n_split, split, t0, t1);

  s16x8 qf[kKS];
  load_row_frags(q + (static_cast<long>(g) * R + min(row, R - 1)) * kC, kh, qf);
  f32x16 acc[kCT];
#pragma unroll
  for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;           // running maximum (log2 units) and this lane's part of the running sum
  need_frags(qf);

  const __amdgpu_buffer_rsrc_t krs = matrix_rsrc(k, total_keys), vrs = matrix_rsrc(v, total_keys);
  const unsigned g_byte = static_cast<unsigned>(g) * static_cast<unsigned>(L) * kRowBytes;       // < 2^32: checked by the host
  const unsigned* bits_o = keybits + static_cast<long>(o) * bits_stride;
  if (t0 < t1) {
    dma_tile(krs, g_byte + static_cast<unsigned>(t0) * kTile, lds, wave, lane);
    dma_tile(vrs, g_byte + static_cast<unsigned>(t0) * kTile, lds + kTile, wave, lane);
  }
  dma_wait();
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const int st = (t - t0) & 1;
    if (t + 1 < t1) {
      unsigned char* nx = lds + (st ^ 1) * 2 * kTile;
      dma_tile(krs, g_byte + static_cast<unsigned>(t + 1) * kTile, nx, wave, lane);
      dma_tile(vrs, g_byte + static_cast<unsigned>(t + 1) * kTile, nx + kTile, wave, lane);
    }
    // ---- S^T[key][row] = K Q^T.  Two accumulators (a chain of MFMAs on ONE accumulator issues every 64 cycles, not 32)
    // and the K fragments fetched six K steps ahead of their use (one wave per SIMD: nobody else hides the LDS latency)
    f32x16 sT;
    {
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
      s16x8 fk[2][6];
#pragma unroll
      for (int e = 0; e < 6; ++e) fk[0][e] = frag_rows<0>(lds, fs, e);
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        if (gq + 1 < 4) {
#pragma unroll
          for (int e = 0; e < 6; ++e) fk[(gq + 1) & 1][e] = frag_rows<0>(lds, fs, 6 * (gq + 1) + e);
        }
#pragma unroll
        for (int e = 0; e < 6; e += 2) {
          s0 = mfma(fk[gq & 1][e], qf[6 * gq + e], s0);
          s1 = mfma(fk[gq & 1][e + 1], qf[6 * gq + e + 1], s1);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sT[r] = s0[r] + s1[r];
    }
    // ---- mask, online softmax (log2 units)
    const unsigned bits = bits_o[t] >> (4 * kh);            // this lane's keys: (r & 3) + 8 (r >> 2) + 4 kh
    float s2[16];
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool masked = (bits >> ((r & 3) + 8 * (r >> 2))) & 1u;
      s2[r] = masked ? -INFINITY : sT[r] * kLog2e;
      mt = fmaxf(mt, s2[r]);
    }
    mt = fmaxf(mt, other_half(mt));
    const float m_new = fmaxf(m_run, mt);
    // Deferred rescaling: the accumulator follows the running maximum only when some row's maximum has grown by more than
    // 2^kDefer since its last rescale; until then P = exp2(S - m_run) may reach 2^kDefer (harmless in fp32 / bf16, and the
    // final division by the row sum undoes it).  Rescaling on every growth ran on most tiles (a wave holds 32 rows: one of
    // them sees a new maximum almost every tile) and cost 576 VALU instructions each time -- 10 per MFMA over the kernel.
    if (__any(m_new > m_run + kDefer)) {
      // rescale the accumulator.  Through explicit accumulator-register moves:
      // written as plain fp32 multiplies, the 192 accumulator values are allocated as arch VGPRs next to the 200 of
      // the softmax / fragment code, and hipcc spills 84 registers around the tile loop.
      const float alpha = m_new == -INFINITY ? 1.f : fast_exp2(m_run - m_new);     // m_run = -inf: exp2(-inf) = 0, acc is 0
#pragma unroll
      for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // in place ("+a"): as separate read / write statements each element became a NEW value inside the branch, and
          // hipcc resolved the join behind it by copying all 192 accumulator registers through VGPRs on EVERY trip
          // (416 v_accvgpr moves per tile: the kernel was VALU-bound at 10 VALU instructions per MFMA)
          float tmp;
          asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\ts_nop 0\n\tv_accvgpr_write_b32 %0, %1"
                       : "+a"(acc[ct][r]), "=&v"(tmp) : "v"(alpha));
        }
      l_run *= alpha;
      m_run = m_new;
    }
    const float m_use = m_run == -INFINITY ? 0.f : m_run;
    float p[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = fast_exp2(s2[r] - m_use);
      l_run += p[r];
    }
    s16x8 pf[2];
    column_to_b_frags(p, pf);
    // ---- O^T[channel][row] += V^T P^T
#pragma unroll
    for (int j = 0; j < 2; ++j)                     // j outermost: consecutive MFMAs write different accumulators
#pragma unroll
      for (int ct = 0; ct < kCT; ++ct) acc[ct] = mfma(frag_cols<kTile>(lds, fs, j, ct), pf[j], acc[ct]);
    frag_shift(fs, st ? -2 * kTile : 2 * kTile);
    dma_wait();
    __syncthreads();
  }

  const float l_tot = l_run + other_half(l_run);
  if (n_split == 1) {
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (row_ok) {
      unsigned short* dst = ctx + (static_cast<long>(g) * R + row) * kC;
#pragma unroll
      for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const u32x2 w{pack_bf16(acc[ct][4 * qd] * inv, acc[ct][4 * qd + 1] * inv),
                        pack_bf16(acc[ct][4 * qd + 2] * inv, acc[ct][4 * qd + 3] * inv)};
          *reinterpret_cast<u32x2*>(dst + 32 * ct + 8 * qd + 4 * kh) = w;
        }
      if (kh == 0) lse[static_cast<long>(g) * R + row] = (m_run + log2f(l_tot)) * kLn2;
    }
  } else if (row_ok) {
    const long prow = (static_cast<long>(g) * n_split + split) * R + row;
    float* dst = pacc + prow * kC;
#pragma unroll
    for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        *reinterpret_cast<float4*>(dst + 32 * ct + 8 * qd + 4 * kh) =
            float4{acc[ct][4 * qd], acc[ct][4 * qd + 1], acc[ct][4 * qd + 2], acc[ct][4 * qd + 3]};
    if (kh == 0) {
      pm[prow] = m_run;
      pl[prow] = l_tot;
    }
  }
}

// partial results of the key splits -> ctx, lse.  One wave per (g, row); lane = 6 channels.
__global__ __launch_bounds__(256) void roi_attn_fwd_combine(const float* __restrict__ pacc, const float* __restrict__ pm,
                                                            const float* __restrict__ pl, unsigned short* __restrict__ ctx,
                                                            float* __restrict__ lse, long n_rows, int R, int n_split) {
  const long gr = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (gr >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const long g = gr / R;
  const int row = static_cast<int>(gr - g * R);
  float m = -INFINITY;
  for (int s = 0; s < n_split; ++s) m = fmaxf(m, pm[(g * n_split + s) * R + row]);
  float l = 0.f, a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < n_split; ++s) {
    const long prow = (g * n_split + s) * R + row;
    const float ms = pm[prow];
    const float w = ms == -INFINITY ? 0.f : fast_exp2(ms - m);
    l += pl[prow] * w;
    const float2* src = reinterpret_cast<const float2*>(pacc + prow * kC + 6 * lane);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float2 x = src[i];
      a[2 * i] += x.x * w;
      a[2 * i + 1] += x.y * w;
    }
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
  unsigned* dst = reinterpret_cast<unsigned*>(ctx + gr * kC + 6 * lane);
#pragma unroll
  for (int i = 0; i < 3; ++i) dst[i] = pack_bf16(a[2 * i] * inv, a[2 * i + 1] * inv);
  if (lane == 0) lse[gr] = (m + log2f(l)) * kLn2;
}

// ---------------------------------------------------------------------------
// backward, query-stationary: dqf = dS k.  grid (n_split, row blocks of 128, G).  Also writes
// dsum[g][row] = rowsum(dctx o ctx) (split 0) for the key-stationary kernel.
// ---------------------------------------------------------------------------
TRANSOAR_ATTN_KERNEL void roi_attn_bwd_q(
    const unsigned short* __restrict__ q, const unsigned short* __restrict__ k, const unsigned short* __restrict__ v,
    const unsigned short* __restrict__ ctx, const unsigned short* __restrict__ dctx, const float* __restrict__ lse,
    const unsigned* __restrict__ keybits, const int* __restrict__ n_tiles, unsigned short* __restrict__ dq,
    float* __restrict__ pdq, float* __restrict__ dsum, int O, int R, int L, long total_keys, int bits_stride, int n_split) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * kTile];
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const int split = blockIdx.x, rb = blockIdx.y, g = blockIdx.z;
  const int o = g % O;
  const int kh = lane >> 5;
  FragBase fs = frag_base(lane);          // fragment bases of the ring stage being read (stage 0 first)
  const int row = rb * 128 + wave * 32 + (lane & 31);
  const bool row_ok = row < R;
  const long grow = static_cast<long>(g) * R + min(row, R - 1);
  int t0, t1;
  split_range(n_tiles[o], n_split, split, t0, t1);

  s16x8 qf[kKS], df[kKS];
  load_row_frags(q + grow * kC, kh, qf);
  load_row_frags(dctx + grow * kC, kh, df);
  // D = rowsum(dctx o ctx): this lane's half of the channels, then the other half's
  float dpart = 0.f;
#pragma unroll
  for (int ks = 0; ks < kKS; ++ks) {
    const u32x4 c4 = *reinterpret_cast<const u32x4*>(ctx + grow * kC + 16 * ks + 8 * kh);
    const u32x4 d4 = __builtin_bit_cast(u32x4, df[ks]);
#pragma unroll
    for (int i = 0; i < 4; ++i) dpart += bf16_lo(c4[i]) * bf16_lo(d4[i]) + bf16_hi(c4[i]) * bf16_hi(d4[i]);
  }
  const float dsum_row = dpart + other_half(dpart);
  if (split == 0 && row_ok && kh == 0) dsum[static_cast<long>(g) * R + row] = dsum_row;
  float lse2 = row_ok ? lse[grow] * kLog2e : INFINITY;       // padding rows: P = 0
  need(lse2);
  need_frags(qf);
  need_frags(df);
  f32x16 acc[kCT];
#pragma unroll
  for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

  const __amdgpu_buffer_rsrc_t krs = matrix_rsrc(k, total_keys), vrs = matrix_rsrc(v, total_keys);
  const unsigned g_byte = static_cast<unsigned>(g) * static_cast<unsigned>(L) * kRowBytes;       // < 2^32: checked by the host
  const unsigned* bits_o = keybits + static_cast<long>(o) * bits_stride;
  if (t0 < t1) {
    dma_tile(krs, g_byte + static_cast<unsigned>(t0) * kTile, lds, wave, lane);
    dma_tile(vrs, g_byte + static_cast<unsigned>(t0) * kTile, lds + kTile, wave, lane);
  }
  dma_wait();
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const int st = (t - t0) & 1;
    if (t + 1 < t1) {
      unsigned char* nx = lds + (st ^ 1) * 2 * kTile;
      dma_tile(krs, g_byte + static_cast<unsigned>(t + 1) * kTile, nx, wave, lane);
      dma_tile(vrs, g_byte + static_cast<unsigned>(t + 1) * kTile, nx + kTile, wave, lane);
    }
    f32x16 sT, dpT;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sT[r] = 0.f; dpT[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {              // the two chains interleaved: consecutive MFMAs on different accumulators
      sT = mfma(frag_rows<0>(lds, fs, ks), qf[ks], sT);
      dpT = mfma(frag_rows<kTile>(lds, fs, ks), df[ks], dpT);
    }
    const unsigned bits = bits_o[t] >> (4 * kh);
    unsigned dpk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float ds2[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int r = 2 * i + e;
        const bool masked = (bits >> ((r & 3) + 8 * (r >> 2))) & 1u;
        const float pr = masked ? 0.f : fast_exp2(sT[r] * kLog2e - lse2);
        ds2[e] = pr * (dpT[r] - dsum_row);
      }
      dpk[i] = pack_bf16(ds2[0], ds2[1]);
    }
    s16x8 dsf[2];
    packed_column_to_b_frags(dpk, dsf);
    // dQ^T[channel][row] += K^T dS^T
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ct = 0; ct < kCT; ++ct) acc[ct] = mfma(frag_cols<0>(lds, fs, j, ct), dsf[j], acc[ct]);
    frag_shift(fs, st ? -2 * kTile : 2 * kTile);
    dma_wait();
    __syncthreads();
  }
  if (!row_ok) return;
  if (n_split == 1) {
    unsigned short* dst = dq + (static_cast<long>(g) * R + row) * kC;
#pragma unroll
    for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const u32x2 w{pack_bf16(acc[ct][4 * qd], acc[ct][4 * qd + 1]), pack_bf16(acc[ct][4 * qd + 2], acc[ct][4 * qd + 3])};
        *reinterpret_cast<u32x2*>(dst + 32 * ct + 8 * qd + 4 * kh) = w;
      }
  } else {
    float* dst = pdq + ((static_cast<long>(g) * n_split + split) * R + row) * kC;
#pragma unroll
    for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        *reinterpret_cast<float4*>(dst + 32 * ct + 8 * qd + 4 * kh) =
            float4{acc[ct][4 * qd], acc[ct][4 * qd + 1], acc[ct][4 * qd + 2], acc[ct][4 * qd + 3]};
  }
}

// sum of the key splits' partial dqf -> bf16.  One thread per 4 channels.
__global__ __launch_bounds__(256) void roi_attn_sum_splits(const float* __restrict__ part, unsigned short* __restrict__ out,
                                                           long n_rows, int R, int n_split) {
  const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;       // (g * R + row) * 96 + c4
  if (i >= n_rows * (kC / 4)) return;
  const long gr = i / (kC / 4);
  const int c4 = static_cast<int>(i - gr * (kC / 4));
  const long g = gr / R;
  const int row = static_cast<int>(gr - g * R);
  float4 a{0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < n_split; ++s) {
    const float4 x = *reinterpret_cast<const float4*>(part + ((g * n_split + s) * R + row) * kC + 4 * c4);
    a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
  }
  *reinterpret_cast<u32x2*>(out + gr * kC + 4 * c4) = u32x2{pack_bf16(a.x, a.y), pack_bf16(a.z, a.w)};
}

// ---------------------------------------------------------------------------
// backward, key-stationary: dtok = dS^T qf + P^T dctx.  grid (key blocks of 128, G); a wave owns 32 keys.
// ---------------------------------------------------------------------------
TRANSOAR_ATTN_KERNEL void roi_attn_bwd_k(
    const unsigned short* __restrict__ q, const unsigned short* __restrict__ k, const unsigned short* __restrict__ v,
    const unsigned short* __restrict__ dctx, const float* __restrict__ lse, const float* __restrict__ dsum,
    const unsigned* __restrict__ keybits, const int* __restrict__ n_tiles, unsigned short* __restrict__ dtok,
    int O, int R, int L, long total_rows, int bits_stride) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * kTile + kMaxRowTiles * 256];
  float* stats = reinterpret_cast<float*>(lds + 4 * kTile);       // [row tile][lse2 x 32 | dsum x 32]: all of the group's rows, loaded once
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const int kb = blockIdx.x, g = blockIdx.y;
  const int o = g % O;
  const int kh = lane >> 5;
  FragBase fs = frag_base(lane);          // fragment bases of the ring stage being read (stage 0 first)
  const int key0 = kb * 128 + wave * 32;
  const int key = key0 + (lane & 31);
  unsigned short* out_g = dtok + static_cast<long>(g) * L * kC;
  if (kb * 4 >= n_tiles[o]) {
    // every key of the block is padding: its token gradient is zero
    for (int i = threadIdx.x; i < 128 * (kC / 8); i += 256) {
      const int kk = kb * 128 + i / (kC / 8);
      if (kk < L) *reinterpret_cast<u32x4*>(out_g + static_cast<long>(kk) * kC + 8 * (i % (kC / 8))) = u32x4{0u, 0u, 0u, 0u};
    }
    return;
  }
  const long gkey = static_cast<long>(g) * L + min(key, L - 1);
  s16x8 kf[kKS], vf[kKS];
  load_row_frags(k + gkey * kC, kh, kf);
  load_row_frags(v + gkey * kC, kh, vf);
  const unsigned bits_w = key0 < L ? keybits[static_cast<long>(o) * bits_stride + (key0 >> 5)] : 0xffffffffu;
  int masked_i = (bits_w >> (lane & 31)) & 1u;
  need(masked_i);
  const bool masked = masked_i != 0;
  need_frags(kf);
  need_frags(vf);
  f32x16 acc[kCT];
#pragma unroll
  for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

  const __amdgpu_buffer_rsrc_t qrs = matrix_rsrc(q, total_rows), drs = matrix_rsrc(dctx, total_rows);
  const unsigned g_byte = static_cast<unsigned>(g) * static_cast<unsigned>(R) * kRowBytes;
  const int n_rt = (R + 31) >> 5;
  // lse (log2 units) and rowsum(dctx o ctx) of every row of the group -> LDS, before the tile loop (a vector memory
  // load inside the loop would make hipcc wait for the loop's DMA with it)
  for (int i = threadIdx.x; i < n_rt * 64; i += 256) {
    const int rr = (i >> 6) * 32 + (i & 31);
    float x;
    if ((i & 32) == 0) x = rr < R ? lse[static_cast<long>(g) * R + rr] * kLog2e : INFINITY;       // padding rows: P = 0
    else x = rr < R ? dsum[static_cast<long>(g) * R + rr] : 0.f;
    stats[i] = x;
  }
  dma_tile(qrs, g_byte, lds, wave, lane);
  dma_tile(drs, g_byte, lds + kTile, wave, lane);
  dma_wait();
  __syncthreads();
  for (int rt = 0; rt < n_rt; ++rt) {
    const int st = rt & 1;
    if (rt + 1 < n_rt) {
      unsigned char* nx = lds + (st ^ 1) * 2 * kTile;
      dma_tile(qrs, g_byte + static_cast<unsigned>(rt + 1) * kTile, nx, wave, lane);
      dma_tile(drs, g_byte + static_cast<unsigned>(rt + 1) * kTile, nx + kTile, wave, lane);
    }
    // S[row][key] = Q K^T and dP[row][key] = dctx V^T: lane = (key, half), register r = row (r & 3) + 8 (r >> 2) + 4 kh
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    // (interleaving the two chains, as the dq kernel does, costs this kernel 39 spilled registers: it is at 512)
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) s = mfma(frag_rows<0>(lds, fs, ks), kf[ks], s);
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) dp = mfma(frag_rows<kTile>(lds, fs, ks), vf[ks], dp);
    unsigned ppk[8], dpk[8];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const float4 l4 = *reinterpret_cast<const float4*>(stats + rt * 64 + 8 * qd + 4 * kh);
      const float4 d4 = *reinterpret_cast<const float4*>(stats + rt * 64 + 32 + 8 * qd + 4 * kh);
      const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
      float pr[4], dsr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * qd + i;
        pr[i] = masked ? 0.f : fast_exp2(s[r] * kLog2e - ls[i]);
        dsr[i] = pr[i] * (dp[r] - dd[i]);
      }
      ppk[2 * qd] = pack_bf16(pr[0], pr[1]); ppk[2 * qd + 1] = pack_bf16(pr[2], pr[3]);
      dpk[2 * qd] = pack_bf16(dsr[0], dsr[1]); dpk[2 * qd + 1] = pack_bf16(dsr[2], dsr[3]);
    }
    s16x8 pf[2], dsf[2];
    packed_column_to_b_frags(ppk, pf);
    packed_column_to_b_frags(dpk, dsf);
    // dTok^T[channel][key] += Q^T dS + dctx^T P
    // (ct outermost: with j outermost -- consecutive MFMAs on different accumulators, as in the other two kernels --
    // hipcc keeps 39 more registers alive than this kernel has)
#pragma unroll
    for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[ct] = mfma(frag_cols<0>(lds, fs, j, ct), dsf[j], acc[ct]);
        acc[ct] = mfma(frag_cols<kTile>(lds, fs, j, ct), pf[j], acc[ct]);
      }
    frag_shift(fs, st ? -2 * kTile : 2 * kTile);
    dma_wait();
    __syncthreads();
  }
  // ---- the wave's 32 x 384 result -> LDS [key][channel] bf16 -> whole 768-byte rows
  unsigned char* ob = lds + wave * kTile;
#pragma unroll
  for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
      // the 48 16-byte pieces of row r sit at piece ^ (r & 15): 768-byte rows put all 16 rows of a store's lane group on the
      // same banks (16-way: 46 % of this kernel's LDS cycles were conflict cycles)
      *reinterpret_cast<u32x2*>(ob + (lane & 31) * kRowBytes + (((4 * ct + qd) ^ (lane & 15)) << 4) + 8 * kh) =
          u32x2{pack_bf16(acc[ct][4 * qd], acc[ct][4 * qd + 1]), pack_bf16(acc[ct][4 * qd + 2], acc[ct][4 * qd + 3])};
  // (a wave's LDS operations execute in order: its reads below see its writes above; no other wave touches this region)
#pragma unroll
  for (int i = 0; i < 24; ++i) {
    const int off = i * 1024 + lane * 16;
    const int kr = ((off >> 8) * 171) >> 9;
    const int piece = (off >> 4) - 48 * kr;
    if (key0 + kr < L)
      *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(out_g + static_cast<long>(key0) * kC) + off) =
          *reinterpret_cast<const u32x4*>(ob + kr * kRowBytes + ((piece ^ (kr & 15)) << 4));
  }
}

// ===========================================================================
// Swin 3-D window attention (transoar/models/backbones/encoder_blocks.py:56-140,
// WindowAttention3D): per (window, head)
//     S = scale q k^T + relative-position bias[head] + shifted-window mask[window]      (n <= 128 tokens, head dim 32)
//     out = softmax(S) v
// Round 3 ran this on torch SDPA with a dense additive mask (the fp32 math path: six fp32 batched GEMMs, softmax,
// isneginf / where / reduce passes over (windows, heads, n, n) tensors).  Head dimension 16 or 32.  Here one workgroup owns a (window, head):
// K and V (128 x 32) sit in LDS, a wave owns 32 query rows and holds all their 128 scores in registers (exact
// softmax, no running maximum), bias rows are read as aligned float4 (the host pads the key axis to 128), the mask is
// one bit per (row, key) ("region labels differ").  The backward kernel is persistent over the windows of a head so
// that the bias gradient accumulates in registers and reaches memory once per workgroup.
// Tiles [rows][32 channels] keep their four 16-byte pieces at piece ^ ((row >> 2) & 3): conflict-free for the
// ds_read_b128 operand fragments, and a transposing read's four rows still tile the 64 banks.
// ===========================================================================
constexpr int kWinN = 128;                 // tokens of a window, padded
constexpr int kWinPPitch = 256;            // bytes per row of the P / dS tiles: no padding, the 8-byte pieces of a row are swizzled (win_p_swz)

// Tiles [128 rows][HD channels] bf16, HD = 16 or 32 (the shipped configurations have 16: 48 / 96 / 192 / 384 channels
// with 3 / 6 / 12 / 24 heads, config/attn_fpn_*: encoder stage k works at the width of stage k - 1's output).  A row's
// 16-byte pieces are XORed with row bits so that the 16 rows of a ds_read_b128 lane group hit 16 different 4-bank slots.
template <int HD> struct WinTile {
  static constexpr int kRow = HD * 2;                  // bytes per row: 32 or 64
  static constexpr int kPieces = HD / 8;               // 2 or 4
  static constexpr int kBytes = kWinN * kRow;
  static __device__ __forceinline__ int off(int row, int piece) {
    const int key = HD == 32 ? (row >> 2) & 3 : (row >> 3) & 1;
    return row * kRow + ((piece ^ key) << 4);
  }
  // rows [r0, r0 + 32) as an MFMA operand [32 rows][16 channels of K step ks] (HD = 16: ks = 0 only)
  static __device__ __forceinline__ s16x8 rows(const unsigned char* tile, int lane, int r0, int ks) {
    return *reinterpret_cast<const s16x8*>(tile + off(r0 + (lane & 31), 2 * ks + (lane >> 5)));
  }
  // the transposed tile as an MFMA operand [32 channels][16 rows R0 + 8 kh .. + 7]; HD = 16: channels 16..31 are zero
  static __device__ __forceinline__ s16x8 cols(const unsigned char* tile, int lane, int R0) {
    const int kh = lane >> 5, r = (lane & 15) >> 2, g = HD == 32 ? (lane >> 4) & 1 : 0, c = lane & 3;
    const int n0 = R0 + 8 * kh + r, n1 = n0 + 4;
    const int p = 2 * g + (c >> 1);
    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off(n0, p) + 8 * (c & 1)));
    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off(n1, p) + 8 * (c & 1)));
    s16x8 f = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    if (HD == 16 && ((lane >> 4) & 1)) f = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    return f;
  }
  // the same in two halves (global -> registers, registers -> LDS): the backward fetches the NEXT window's tiles while it
  // works on this one
  struct Regs { u32x4 a, b; };
  static __device__ __forceinline__ Regs fetch(const unsigned short* __restrict__ src, long tok_elems, int n) {
    const int row = (threadIdx.x & 255) >> 1, half = threadIdx.x & 1;          // (a 512-thread workgroup: each half stages its own tiles)
    Regs r{u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
    if (row < n) {
      if constexpr (HD == 32) {
        const u32x4* g = reinterpret_cast<const u32x4*>(src + row * tok_elems + 16 * half);
        r.a = g[0];
        r.b = g[1];
      } else {
        r.a = *reinterpret_cast<const u32x4*>(src + row * tok_elems + 8 * half);
      }
    }
    return r;
  }
  static __device__ __forceinline__ void put(unsigned char* tile, const Regs& r) {
    const int row = (threadIdx.x & 255) >> 1, half = threadIdx.x & 1;
    if constexpr (HD == 32) {
      *reinterpret_cast<u32x4*>(tile + off(row, 2 * half)) = r.a;
      *reinterpret_cast<u32x4*>(tile + off(row, 2 * half + 1)) = r.b;
    } else {
      *reinterpret_cast<u32x4*>(tile + off(row, half)) = r.a;
    }
  }
};
// The P / dS tiles [128 rows][128 keys] bf16 keep the 32 8-byte pieces of row i at piece ^ win_p_swz(i) (round 5; they had a
// 320-byte pitch: rows i and i + 2 of a ds_write_b64 lane group shared their banks, 71 % of the kernel's LDS cycles were
// conflict cycles -- and 96 KiB per workgroup).  Bits 4:3 of the swizzle are the row's two low bits: the four rows of a
// transposing read (4 rows x 8 pieces per 32-lane group) land in four different 64-byte windows and tile the 64 read
// banks; its low four bits take all 16 values over the 16 rows of a store's lane group, which then covers the 32 store
// banks once.
__device__ __forceinline__ int win_p_swz(int row) { return ((row & 3) << 3) | (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
// a [row][key] tile (pitch kWinPPitch) as the MFMA B operand [k = rows R0 + 8 kh .. + 7][n = key K0 + (lane & 31)]
__device__ __forceinline__ s16x8 wp_frag(const unsigned char* tile, int lane, int R0, int K0) {
  const int kh = lane >> 5, r = (lane & 15) >> 2, g = (lane >> 4) & 1, c = lane & 3;
  const int row = R0 + 8 * kh + r, piece = (K0 >> 2) + 4 * g + c;            // 8-byte piece of the row, swizzled like the stores
  const unsigned char* a0 = tile + row * kWinPPitch + ((piece ^ win_p_swz(row)) << 3);
  const unsigned char* a1 = tile + (row + 4) * kWinPPitch + ((piece ^ win_p_swz(row + 4)) << 3);
  const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
  const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
  return __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Accumulator the score MFMAs of key tile t start from (round 5): bias / scale, so that scale2 * (k.q + init) is the
// score in log2 units with ONE fma per entry -- and -inf where the key is padding (K's padding rows are zero: the entry
// stays -inf).  lane = (row i, half kh), entry r = key 32 t + (r & 3) + 8 (r >> 2) + 4 kh.
__device__ __forceinline__ void win_bias_init(const float* __restrict__ bias_row, int t, int kh, int n, float inv_scale, bool row_ok,
                                              f32x16& c) {
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const int key0 = 32 * t + 8 * qd + 4 * kh;
    float4 b4{0.f, 0.f, 0.f, 0.f};
    if (row_ok) b4 = *reinterpret_cast<const float4*>(bias_row + key0);
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) c[4 * qd + e] = bb[e] * inv_scale;
    if (32 * t + 8 * qd + 8 > n) {                       // wave-uniform: only the group that holds key n .. pays for the test
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (key0 + e >= n) c[4 * qd + e] = -INFINITY;
    }
  }
}
// ... plus the shifted-window mask (-100 where the region labels differ) of a tile that has any: neg = -100 / scale
__device__ __forceinline__ void win_mask_apply(unsigned mask_word, int kh, float neg, f32x16& c) {
#pragma unroll
  for (int qd = 0; qd < 4; ++qd)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if ((mask_word >> (8 * qd + 4 * kh + e)) & 1u) c[4 * qd + e] += neg;
}
__device__ __forceinline__ bool wave_any(bool v) { return __builtin_amdgcn_ballot_w64(v) != 0; }
// the wave's 32 x HD result (channel (r & 3) + 8 (r >> 2) + 4 kh of row / key `lane & 31`) -> 8-byte pieces of a token row
template <int HD>
__device__ __forceinline__ void win_store(unsigned short* __restrict__ dst, const f32x16& acc, int kh) {
#pragma unroll
  for (int qd = 0; qd < HD / 8; ++qd)
    *reinterpret_cast<u32x2*>(dst + 8 * qd + 4 * kh) =
        u32x2{pack_bf16(acc[4 * qd], acc[4 * qd + 1]), pack_bf16(acc[4 * qd + 2], acc[4 * qd + 3])};
}

// forward: grid (persistent workgroups, heads); workgroup x walks the windows x, x + gridDim.x, ... of its head (round 5:
// as one workgroup per (window, head) a wave lived 25 000 cycles and waited 78 % of them -- three dependent round trips
// (K / V / q, bias, store) with three short workgroups per CU to cover them).  Now the next window's K, V, q and mask
// bits are requested before this window's arithmetic, K / V tiles are double-buffered in LDS (one barrier per
// window), and the head's bias rows -- the same for every window -- stay in registers.
template <int HD>
__global__ __launch_bounds__(256, 2) void win_attn_fwd(
    const unsigned short* __restrict__ qkv, const float* __restrict__ bias, const unsigned* __restrict__ maskbits,
    unsigned short* __restrict__ out, float* __restrict__ lse2, int n, int heads, int n_win, int windows, float scale) {
  using T = WinTile<HD>;
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * T::kBytes];         // (K, V) x 2 stages
  const int head = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const int kh = lane >> 5;
  const long tok = 3L * heads * HD;                                       // elements per token of qkv
  const int i = wave * 32 + (lane & 31);
  const bool row_ok = i < n;
  const bool wave_live = wave * 32 < n;                 // a wave without a live row (n <= 96) only helps to stage the tiles
  const float scale2 = scale * kLog2e, inv_scale = 1.f / scale, neg = -100.f * inv_scale;
  const float* bias_row = bias + (static_cast<long>(head) * n + (row_ok ? i : 0)) * kWinN;
  f32x16 cb[4];                                         // bias / scale, -inf at padding keys: what the score MFMAs start from
#pragma unroll
  for (int t = 0; t < 4; ++t) win_bias_init(bias_row, t, kh, n, inv_scale, row_ok, cb[t]);

  struct WinPre {
    typename T::Regs k, v;
    u32x4 q[HD / 16], mw;
  };
  auto fetch_window = [&](int w) {
    WinPre pf;
    const unsigned short* base = qkv + static_cast<long>(w) * n * tok + head * HD;
    pf.k = T::fetch(base + heads * HD, tok, n);
    pf.v = T::fetch(base + 2 * heads * HD, tok, n);
    pf.mw = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
      pf.q[ks] = u32x4{0u, 0u, 0u, 0u};
      if (row_ok) pf.q[ks] = *reinterpret_cast<const u32x4*>(base + i * tok + 16 * ks + 8 * kh);
    }
    if (maskbits != nullptr && row_ok) pf.mw = *reinterpret_cast<const u32x4*>(maskbits + (static_cast<long>(w % n_win) * n + i) * 4);
    return pf;
  };
  WinPre pre;
  if (static_cast<int>(blockIdx.x) < windows) pre = fetch_window(blockIdx.x);
  int stage = 0;
  for (int w = blockIdx.x; w < windows; w += gridDim.x, stage ^= 1) {
    unsigned char* kt = lds + stage * 2 * T::kBytes;
    unsigned char* vt = kt + T::kBytes;
    T::put(kt, pre.k);
    T::put(vt, pre.v);
    s16x8 qf[HD / 16];
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) qf[ks] = __builtin_bit_cast(s16x8, pre.q[ks]);
    const u32x4 mw = pre.mw;
    // one barrier per window: a wave that runs ahead writes the OTHER stage, and cannot reach this one again before every
    // wave has passed the next barrier, i.e. has finished reading it
    __syncthreads();
    if (w + static_cast<int>(gridDim.x) < windows) pre = fetch_window(w + gridDim.x);
    if (!wave_live) continue;

    f32x16 sT[4];                                        // raw scores k.q + bias / scale: log2 units = scale2 * sT (scale > 0)
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      sT[t] = cb[t];
      if (maskbits != nullptr && wave_any(mw[t] != 0u)) win_mask_apply(mw[t], kh, neg, sT[t]);
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) sT[t] = mfma(T::rows(kt, lane, 32 * t, ks), qf[ks], sT[t]);
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, sT[t][r]);
    }
    m = fmaxf(m, other_half(m));
    const float m2 = m * scale2;
    float l = 0.f;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // P stays unnormalised (<= 1) on its way through the matrix cores; the row's 1 / l scales the 32 x HD result
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float p[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = fast_exp2(__builtin_fmaf(sT[t][r], scale2, -m2));
        l += p[r];
      }
      s16x8 pf[2];
      column_to_b_frags(p, pf);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc = mfma(T::cols(vt, lane, 32 * t + 16 * j), pf[j], acc);
    }
    l += other_half(l);
    const float inv = 1.f / l;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] *= inv;
    if (row_ok) {
      win_store<HD>(out + (static_cast<long>(w) * n + i) * (heads * HD) + head * HD, acc, kh);
      if (kh == 0) lse2[(static_cast<long>(w) * heads + head) * n + i] = m2 + log2f(l);
    }
  }
}

// backward: grid (persistent workgroups, heads); workgroup x walks the windows x, x + gridDim.x, ... of its head
template <int HD>
__global__ __launch_bounds__(256) void win_attn_bwd(
    const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ out, const unsigned short* __restrict__ dout,
    const float* __restrict__ lse2, const float* __restrict__ bias, const unsigned* __restrict__ maskbits,
    unsigned short* __restrict__ dqkv, float* __restrict__ dbias, int n, int heads, int n_win, int windows, float scale) {
  using T = WinTile<HD>;
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * T::kBytes + 2 * kWinN * kWinPPitch];
  unsigned char* kt = lds;
  unsigned char* vt = lds + T::kBytes;
  unsigned char* qt = lds + 2 * T::kBytes;
  unsigned char* dt = lds + 3 * T::kBytes;
  unsigned char* pt = lds + 4 * T::kBytes;                      // P   [row][key] bf16
  unsigned char* st = pt + kWinN * kWinPPitch;                  // scale * dS [row][key] bf16
  const int head = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const int kh = lane >> 5;
  const long tok = 3L * heads * HD;
  const int C = heads * HD;
  const int i = wave * 32 + (lane & 31);
  const bool row_ok = i < n;
  const float scale2 = scale * kLog2e;
  const float* bias_row = bias + (static_cast<long>(head) * n + (row_ok ? i : 0)) * kWinN;
  const float inv_scale = 1.f / scale, neg = -100.f * inv_scale;
  float db[4][16];                                               // scale * bias gradient of (row i, this lane's 64 keys), summed over the windows
  // bias / scale (-inf at padding keys): what the score MFMAs start from, the same for every window.  (Two 80-KiB workgroups
  // per CU would fit the LDS since the P / dS tiles lost their padding, but not the register file: at 256 registers --
  // without these 64 -- hipcc spills 81 and the kernel takes 1.8 x as long.)
  constexpr bool kKeepBias = true;
  f32x16 cb[kKeepBias ? 4 : 1];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if constexpr (kKeepBias) win_bias_init(bias_row, t, kh, n, inv_scale, row_ok, cb[t]);
#pragma unroll
    for (int r = 0; r < 16; ++r) db[t][r] = 0.f;
  }

  // Everything a window needs from global memory is requested one window ahead (round 5): the four operand tiles as
  // 16 / 32 bytes per thread, and per row the out / dout pieces of D = rowsum(dout o out), the log-sum-exp and the mask
  // bits.  With one 96-KiB workgroup per CU nothing else covers that latency -- it was a quarter of a window's time.
  struct WinPre {
    typename T::Regs q, k, v, d;
    u32x4 o4[HD / 16], d4[HD / 16], mw;
    float lse;
  };
  auto fetch_window = [&](int w) {
    WinPre pf;
    const unsigned short* base = qkv + static_cast<long>(w) * n * tok + head * HD;
    pf.q = T::fetch(base, tok, n);
    pf.k = T::fetch(base + heads * HD, tok, n);
    pf.v = T::fetch(base + 2 * heads * HD, tok, n);
    pf.d = T::fetch(dout + static_cast<long>(w) * n * C + head * HD, C, n);
    pf.mw = u32x4{0u, 0u, 0u, 0u};
    pf.lse = INFINITY;
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) { pf.o4[ks] = u32x4{0u, 0u, 0u, 0u}; pf.d4[ks] = u32x4{0u, 0u, 0u, 0u}; }
    if (row_ok) {
      const unsigned short* orow = out + (static_cast<long>(w) * n + i) * C + head * HD;
      const unsigned short* drow = dout + (static_cast<long>(w) * n + i) * C + head * HD;
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) {
        pf.o4[ks] = *reinterpret_cast<const u32x4*>(orow + 16 * ks + 8 * kh);
        pf.d4[ks] = *reinterpret_cast<const u32x4*>(drow + 16 * ks + 8 * kh);
      }
      pf.lse = lse2[(static_cast<long>(w) * heads + head) * n + i];
      if (maskbits != nullptr) pf.mw = *reinterpret_cast<const u32x4*>(maskbits + (static_cast<long>(w % n_win) * n + i) * 4);
    }
    return pf;
  };
  WinPre pre;
  if (static_cast<int>(blockIdx.x) < windows) pre = fetch_window(blockIdx.x);
  for (int w = blockIdx.x; w < windows; w += gridDim.x) {
    T::put(qt, pre.q);
    T::put(kt, pre.k);
    T::put(vt, pre.v);
    T::put(dt, pre.d);
    // D = rowsum(dout o out) of row i (this lane: half of the head's channels), the row's log-sum-exp, its mask bits
    float dpart = 0.f;
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        dpart += bf16_lo(pre.o4[ks][e]) * bf16_lo(pre.d4[ks][e]) + bf16_hi(pre.o4[ks][e]) * bf16_hi(pre.d4[ks][e]);
    const float dsum_s = (dpart + other_half(dpart)) * scale;
    const float lse_i = pre.lse;
    const u32x4 mw = pre.mw;
    __syncthreads();
    if (w + static_cast<int>(gridDim.x) < windows) pre = fetch_window(w + gridDim.x);

    // ---- row side: P, dS of the wave's 32 rows; dq; P and scale dS -> LDS
    s16x8 qf[HD / 16], df[HD / 16];
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
      qf[ks] = T::rows(qt, lane, wave * 32, ks);
      df[ks] = T::rows(dt, lane, wave * 32, ks);
    }
    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x16 sT, dpT;
      if constexpr (kKeepBias) sT = cb[t];
      else win_bias_init(bias_row, t, kh, n, inv_scale, row_ok, sT);
      if (maskbits != nullptr && wave_any(mw[t] != 0u)) win_mask_apply(mw[t], kh, neg, sT);
#pragma unroll
      for (int r = 0; r < 16; ++r) dpT[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) {
        sT = mfma(T::rows(kt, lane, 32 * t, ks), qf[ks], sT);
        dpT = mfma(T::rows(vt, lane, 32 * t, ks), df[ks], dpT);
      }
      unsigned ppk[8], dpk[8];
#pragma unroll
      for (int e2 = 0; e2 < 8; ++e2) {
        float pr[2], dsr[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int r = 2 * e2 + e;
          pr[e] = fast_exp2(__builtin_fmaf(sT[r], scale2, -lse_i));      // padding rows: lse = +inf -> 0; keys >= n: sT = -inf -> 0
          dsr[e] = pr[e] * __builtin_fmaf(dpT[r], scale, -dsum_s);       // scale * dS = scale * P (dP - D)
          db[t][r] += dsr[e];
        }
        ppk[e2] = pack_bf16(pr[0], pr[1]);
        dpk[e2] = pack_bf16(dsr[0], dsr[1]);
      }
      // entries 4 qd .. 4 qd + 3 = keys 32 t + 8 qd + 4 kh .. + 3: 8 bytes of the row's P / dS line
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int off = i * kWinPPitch + (((8 * t + 2 * qd + kh) ^ win_p_swz(i)) << 3);
        *reinterpret_cast<u32x2*>(pt + off) = u32x2{ppk[2 * qd], ppk[2 * qd + 1]};
        *reinterpret_cast<u32x2*>(st + off) = u32x2{dpk[2 * qd], dpk[2 * qd + 1]};
      }
      s16x8 dsf[2];
      packed_column_to_b_frags(dpk, dsf);
      // dQ^T[channel][row] += K^T (scale dS)^T
#pragma unroll
      for (int j = 0; j < 2; ++j) dq = mfma(T::cols(kt, lane, 32 * t + 16 * j), dsf[j], dq);
    }
    if (row_ok) win_store<HD>(dqkv + (static_cast<long>(w) * n + i) * tok + head * HD, dq, kh);
    __syncthreads();

    // ---- key side: the wave's 32 keys.  dV^T[channel][key] = dout^T P,  dK^T[channel][key] = q^T (scale dS)
    f32x16 dv, dk;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dv[r] = 0.f; dk[r] = 0.f; }
#pragma unroll
    for (int jr = 0; jr < 8; ++jr) {
      dv = mfma(T::cols(dt, lane, 16 * jr), wp_frag(pt, lane, 16 * jr, wave * 32), dv);
      dk = mfma(T::cols(qt, lane, 16 * jr), wp_frag(st, lane, 16 * jr, wave * 32), dk);
    }
    if (row_ok) {                                                   // here i is the lane's KEY
      unsigned short* dst = dqkv + (static_cast<long>(w) * n + i) * tok + head * HD;
      win_store<HD>(dst + heads * HD, dk, kh);
      win_store<HD>(dst + 2 * heads * HD, dv, kh);
    }
    __syncthreads();
  }
  // ---- the bias gradient of this workgroup's windows: one atomic per (row, key)
  if (row_ok) {
    float* drow = dbias + (static_cast<long>(head) * n + i) * kWinN;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (key < n) atomicAdd(drow + key, db[t][r] * inv_scale);
      }
  }
}

// The same backward with EIGHT waves per workgroup (round 5): two waves per SIMD of the one workgroup a CU holds, so that one
// wave's exponentials run under the other's MFMAs and LDS round trips.  The row side splits the keys (wave = row tile x key
// half: two of the four 32-key tiles each -- half the bias-gradient and bias registers), the key side splits the rows
// (wave = key tile x row half); the half-sums of dq and of (dv, dk) meet through LDS across the barrier that follows them
// anyway.  Staging: threads 0..255 fetch / put q and k, threads 256..511 v and dout.
template <int HD>
__global__ __launch_bounds__(512) void win_attn_bwd8(
    const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ out, const unsigned short* __restrict__ dout,
    const float* __restrict__ lse2, const float* __restrict__ bias, const unsigned* __restrict__ maskbits,
    unsigned short* __restrict__ dqkv, float* __restrict__ dbias, int n, int heads, int n_win, int windows, float scale) {
  using T = WinTile<HD>;
  constexpr int kXq = 4 * 64 * 16 * 4, kXkv = 4 * 64 * 32 * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * T::kBytes + 2 * kWinN * kWinPPitch + kXq + kXkv];
  unsigned char* kt = lds;
  unsigned char* vt = lds + T::kBytes;
  unsigned char* qt = lds + 2 * T::kBytes;
  unsigned char* dt = lds + 3 * T::kBytes;
  unsigned char* pt = lds + 4 * T::kBytes;                      // P   [row][key] bf16
  unsigned char* st = pt + kWinN * kWinPPitch;                  // scale * dS [row][key] bf16
  float* xq = reinterpret_cast<float*>(st + kWinN * kWinPPitch);          // dq of the upper key half: [row tile][lane][16]
  float* xkv = xq + kXq / 4;                                              // dv | dk of the upper row half: [key tile][lane][32]
  const int head = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const int rt = wave & 3, hs = wave >> 2;                      // row tile (key tile on the key side); key half (row half)
  const int kh = lane >> 5;
  const long tok = 3L * heads * HD;
  const int C = heads * HD;
  const int i = rt * 32 + (lane & 31);
  const bool row_ok = i < n;
  const float scale2 = scale * kLog2e;
  const float* bias_row = bias + (static_cast<long>(head) * n + (row_ok ? i : 0)) * kWinN;
  const float inv_scale = 1.f / scale, neg = -100.f * inv_scale;
  float db[2][16];
  f32x16 cb[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    win_bias_init(bias_row, 2 * hs + tt, kh, n, inv_scale, row_ok, cb[tt]);
#pragma unroll
    for (int r = 0; r < 16; ++r) db[tt][r] = 0.f;
  }
  struct WinPre {
    typename T::Regs a, b;            // hs 0: q, k;  hs 1: v, dout
    u32x4 o4[HD / 16], d4[HD / 16];
    unsigned mw[2];
    float lse;
  };
  auto fetch_window = [&](int w) {
    WinPre pf;
    const unsigned short* base = qkv + static_cast<long>(w) * n * tok + head * HD;
    if (hs == 0) {
      pf.a = T::fetch(base, tok, n);
      pf.b = T::fetch(base + heads * HD, tok, n);
    } else {
      pf.a = T::fetch(base + 2 * heads * HD, tok, n);
      pf.b = T::fetch(dout + static_cast<long>(w) * n * C + head * HD, C, n);
    }
    pf.mw[0] = pf.mw[1] = 0u;
    pf.lse = INFINITY;
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) { pf.o4[ks] = u32x4{0u, 0u, 0u, 0u}; pf.d4[ks] = u32x4{0u, 0u, 0u, 0u}; }
    if (row_ok) {
      const unsigned short* orow = out + (static_cast<long>(w) * n + i) * C + head * HD;
      const unsigned short* drow = dout + (static_cast<long>(w) * n + i) * C + head * HD;
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) {
        pf.o4[ks] = *reinterpret_cast<const u32x4*>(orow + 16 * ks + 8 * kh);
        pf.d4[ks] = *reinterpret_cast<const u32x4*>(drow + 16 * ks + 8 * kh);
      }
      pf.lse = lse2[(static_cast<long>(w) * heads + head) * n + i];
      if (maskbits != nullptr) {
        const uint2 m2 = *reinterpret_cast<const uint2*>(maskbits + (static_cast<long>(w % n_win) * n + i) * 4 + 2 * hs);
        pf.mw[0] = m2.x; pf.mw[1] = m2.y;
      }
    }
    return pf;
  };
  WinPre pre;
  if (static_cast<int>(blockIdx.x) < windows) pre = fetch_window(blockIdx.x);
  for (int w = blockIdx.x; w < windows; w += gridDim.x) {
    if (hs == 0) { T::put(qt, pre.a); T::put(kt, pre.b); }
    else { T::put(vt, pre.a); T::put(dt, pre.b); }
    float dpart = 0.f;
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        dpart += bf16_lo(pre.o4[ks][e]) * bf16_lo(pre.d4[ks][e]) + bf16_hi(pre.o4[ks][e]) * bf16_hi(pre.d4[ks][e]);
    const float dsum_s = (dpart + other_half(dpart)) * scale;
    const float lse_i = pre.lse;
    const unsigned mw[2] = {pre.mw[0], pre.mw[1]};
    __syncthreads();
    if (w + static_cast<int>(gridDim.x) < windows) pre = fetch_window(w + gridDim.x);

    // ---- row side: P, dS of the wave's 32 rows x 64 keys; its half of dq; P and scale dS -> LDS
    s16x8 qf[HD / 16], df[HD / 16];
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
      qf[ks] = T::rows(qt, lane, rt * 32, ks);
      df[ks] = T::rows(dt, lane, rt * 32, ks);
    }
    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int t = 2 * hs + tt;
      f32x16 sT = cb[tt], dpT;
      if (maskbits != nullptr && wave_any(mw[tt] != 0u)) win_mask_apply(mw[tt], kh, neg, sT);
#pragma unroll
      for (int r = 0; r < 16; ++r) dpT[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) {
        sT = mfma(T::rows(kt, lane, 32 * t, ks), qf[ks], sT);
        dpT = mfma(T::rows(vt, lane, 32 * t, ks), df[ks], dpT);
      }
      unsigned ppk[8], dpk[8];
#pragma unroll
      for (int e2 = 0; e2 < 8; ++e2) {
        float pr[2], dsr[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int r = 2 * e2 + e;
          pr[e] = fast_exp2(__builtin_fmaf(sT[r], scale2, -lse_i));
          dsr[e] = pr[e] * __builtin_fmaf(dpT[r], scale, -dsum_s);
          db[tt][r] += dsr[e];
        }
        ppk[e2] = pack_bf16(pr[0], pr[1]);
        dpk[e2] = pack_bf16(dsr[0], dsr[1]);
      }
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int off = i * kWinPPitch + (((8 * t + 2 * qd + kh) ^ win_p_swz(i)) << 3);
        *reinterpret_cast<u32x2*>(pt + off) = u32x2{ppk[2 * qd], ppk[2 * qd + 1]};
        *reinterpret_cast<u32x2*>(st + off) = u32x2{dpk[2 * qd], dpk[2 * qd + 1]};
      }
      s16x8 dsf[2];
      packed_column_to_b_frags(dpk, dsf);
#pragma unroll
      for (int j = 0; j < 2; ++j) dq = mfma(T::cols(kt, lane, 32 * t + 16 * j), dsf[j], dq);
    }
    // (piece-major, lane-minor: the 8 lanes of a 16-byte access group touch 128 contiguous bytes; lane-major cost 58 % of the
    // kernel's LDS cycles in bank conflicts)
    float4* xq4 = reinterpret_cast<float4*>(xq) + rt * 256 + lane;
    if (hs == 1) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) xq4[qd * 64] = float4{dq[4 * qd], dq[4 * qd + 1], dq[4 * qd + 2], dq[4 * qd + 3]};
    }
    __syncthreads();
    if (hs == 0) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 o = xq4[qd * 64];
        dq[4 * qd] += o.x; dq[4 * qd + 1] += o.y; dq[4 * qd + 2] += o.z; dq[4 * qd + 3] += o.w;
      }
      if (row_ok) win_store<HD>(dqkv + (static_cast<long>(w) * n + i) * tok + head * HD, dq, kh);
    }

    // ---- key side: the wave's 32 keys x 64 rows (rt is the key tile here, hs the row half)
    f32x16 dv, dk;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dv[r] = 0.f; dk[r] = 0.f; }
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
      const int jr = 4 * hs + j4;
      dv = mfma(T::cols(dt, lane, 16 * jr), wp_frag(pt, lane, 16 * jr, rt * 32), dv);
      dk = mfma(T::cols(qt, lane, 16 * jr), wp_frag(st, lane, 16 * jr, rt * 32), dk);
    }
    float4* xkv4 = reinterpret_cast<float4*>(xkv) + rt * 512 + lane;
    if (hs == 1) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        xkv4[qd * 64] = float4{dv[4 * qd], dv[4 * qd + 1], dv[4 * qd + 2], dv[4 * qd + 3]};
        xkv4[(4 + qd) * 64] = float4{dk[4 * qd], dk[4 * qd + 1], dk[4 * qd + 2], dk[4 * qd + 3]};
      }
    }
    __syncthreads();
    if (hs == 0) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 a = xkv4[qd * 64], b = xkv4[(4 + qd) * 64];
        dv[4 * qd] += a.x; dv[4 * qd + 1] += a.y; dv[4 * qd + 2] += a.z; dv[4 * qd + 3] += a.w;
        dk[4 * qd] += b.x; dk[4 * qd + 1] += b.y; dk[4 * qd + 2] += b.z; dk[4 * qd + 3] += b.w;
      }
      if (row_ok) {                                                   // here i is the lane's KEY
        unsigned short* dst = dqkv + (static_cast<long>(w) * n + i) * tok + head * HD;
        win_store<HD>(dst + heads * HD, dk, kh);
        win_store<HD>(dst + 2 * heads * HD, dv, kh);
      }
    }
    // (no barrier here: the next window's tiles are put into kt .. dt, which every wave finished reading before the barrier
    // above; pt / st are rewritten only after the next window's first barrier; xq / xkv only after its second / third)
  }
  if (row_ok) {
    float* drow = dbias + (static_cast<long>(head) * n + i) * kWinN;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * (2 * hs + tt) + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (key < n) atomicAdd(drow + key, db[tt][r] * inv_scale);
      }
  }
}

}  // namespace transoar

using namespace transoar;

static inline size_t align256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

extern "C" size_t transoar_roi_attn_workspace_bytes(int G, int R, int n_split) {
  if (G <= 0 || R <= 0 || n_split < 1) return 0;
  const size_t rows = static_cast<size_t>(G) * R;
  size_t bytes = align256(rows * sizeof(float));                                  // dsum
  if (n_split > 1) bytes += align256(rows * n_split * kC * sizeof(float)) + 2 * align256(rows * n_split * sizeof(float));
  return bytes;
}

static int check_common(const void* q, const void* k, const void* v, const void* bits, const void* nt, int G, int O, int R,
                        long L, int C, int n_split) {
  if (!q || !k || !v || !bits || !nt) return TRANSOAR_ATTN_ERR_NULL;
  if (C != kC || G <= 0 || O <= 0 || G % O != 0 || R <= 0 || R > 32 * kMaxRowTiles || L <= 0 || n_split < 1 || n_split > 64) return TRANSOAR_ATTN_ERR_DIM;
  // tiles are addressed with 32-bit byte offsets into k / v / q / dctx
  if ((static_cast<long>(G) * L + 64) * kRowBytes >= 0x7fffffffL || (static_cast<long>(G) * R + 64) * kRowBytes >= 0x7fffffffL || G >= 65536)
    return TRANSOAR_ATTN_ERR_DIM;
  return TRANSOAR_ATTN_OK;
}

extern "C" int transoar_roi_attn_forward(const void* q, const void* k, const void* v, const unsigned* keybits, const int* n_tiles,
                                         void* ctx, float* lse, void* workspace, size_t workspace_bytes, int G, int O, int R, long L,
                                         int C, int n_split, void* hip_stream) {
  int rc = check_common(q, k, v, keybits, n_tiles, G, O, R, L, C, n_split);
  if (rc != TRANSOAR_ATTN_OK) return rc;
  if (!ctx || !lse) return TRANSOAR_ATTN_ERR_NULL;
  if (workspace_bytes < transoar_roi_attn_workspace_bytes(G, R, n_split) || (n_split > 1 && !workspace)) return TRANSOAR_ATTN_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const size_t rows = static_cast<size_t>(G) * R;
  char* ws = static_cast<char*>(workspace);
  float* pacc = nullptr; float* pm = nullptr; float* pl = nullptr;
  if (n_split > 1) {
    pacc = reinterpret_cast<float*>(ws + align256(rows * sizeof(float)));
    pm = reinterpret_cast<float*>(reinterpret_cast<char*>(pacc) + align256(rows * n_split * kC * sizeof(float)));
    pl = reinterpret_cast<float*>(reinterpret_cast<char*>(pm) + align256(rows * n_split * sizeof(float)));
  }
  const int bits_stride = static_cast<int>((L + 31) / 32);
  const dim3 grid(n_split, (R + 127) / 128, G);
  hipLaunchKernelGGL(roi_attn_fwd, grid, dim3(256), 0, st, static_cast<const unsigned short*>(q), static_cast<const unsigned short*>(k),
                     static_cast<const unsigned short*>(v), keybits, n_tiles, static_cast<unsigned short*>(ctx), lse, pacc, pm, pl, O, R,
                     static_cast<int>(L), static_cast<long>(G) * L, bits_stride, n_split);
  if (n_split > 1)
    hipLaunchKernelGGL(roi_attn_fwd_combine, dim3(static_cast<unsigned>((rows + 3) / 4)), dim3(256), 0, st, pacc, pm, pl,
                       static_cast<unsigned short*>(ctx), lse, static_cast<long>(rows), R, n_split);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_roi_attn_backward(const void* q, const void* k, const void* v, const void* ctx, const void* dctx, const float* lse,
                                          const unsigned* keybits, const int* n_tiles, void* dq, void* dtok, void* workspace,
                                          size_t workspace_bytes, int G, int O, int R, long L, int C, int n_split, void* hip_stream) {
  int rc = check_common(q, k, v, keybits, n_tiles, G, O, R, L, C, n_split);
  if (rc != TRANSOAR_ATTN_OK) return rc;
  if (!ctx || !dctx || !lse || !dq || !dtok || !workspace) return TRANSOAR_ATTN_ERR_NULL;
  if (workspace_bytes < transoar_roi_attn_workspace_bytes(G, R, n_split)) return TRANSOAR_ATTN_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const size_t rows = static_cast<size_t>(G) * R;
  char* ws = static_cast<char*>(workspace);
  float* dsum = reinterpret_cast<float*>(ws);
  float* pdq = n_split > 1 ? reinterpret_cast<float*>(ws + align256(rows * sizeof(float))) : nullptr;
  const int bits_stride = static_cast<int>((L + 31) / 32);
  auto qs = static_cast<const unsigned short*>(q);
  auto ks = static_cast<const unsigned short*>(k);
  auto vs = static_cast<const unsigned short*>(v);
  auto ds = static_cast<const unsigned short*>(dctx);
  hipLaunchKernelGGL(roi_attn_bwd_q, dim3(n_split, (R + 127) / 128, G), dim3(256), 0, st, qs, ks, vs,
                     static_cast<const unsigned short*>(ctx), ds, lse, keybits, n_tiles, static_cast<unsigned short*>(dq), pdq, dsum, O, R,
                     static_cast<int>(L), static_cast<long>(G) * L, bits_stride, n_split);
  if (n_split > 1)
    hipLaunchKernelGGL(roi_attn_sum_splits, dim3(static_cast<unsigned>((rows * (kC / 4) + 255) / 256)), dim3(256), 0, st, pdq,
                       static_cast<unsigned short*>(dq), static_cast<long>(rows), R, n_split);
  hipLaunchKernelGGL(roi_attn_bwd_k, dim3(static_cast<unsigned>((L + 127) / 128), G), dim3(256), 0, st, qs, ks, vs, ds, lse, dsum, keybits,
                     n_tiles, static_cast<unsigned short*>(dtok), O, R, static_cast<int>(L), static_cast<long>(G) * R, bits_stride);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_win_attn_forward(const void* qkv, const float* bias, const unsigned* maskbits, void* out, float* lse2, int windows,
                                         int n_win, int n, int heads, int head_dim, float scale, void* hip_stream) {
  if (!qkv || !bias || !out || !lse2) return TRANSOAR_ATTN_ERR_NULL;
  if ((head_dim != 16 && head_dim != 32) || n <= 0 || n > kWinN || heads <= 0 || heads > 65535 || windows <= 0 || n_win <= 0 || !(scale > 0.f)) return TRANSOAR_ATTN_ERR_DIM;
  auto qs = static_cast<const unsigned short*>(qkv);
  auto os = static_cast<unsigned short*>(out);
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  // two resident workgroups per CU (<= 256 registers, 16 / 32 KiB of LDS), the windows of a head dealt round-robin
  const int per_head = std::max(1, std::min(windows, 512 / std::min(heads, 512)));
  if (head_dim == 32) hipLaunchKernelGGL(win_attn_fwd<32>, dim3(per_head, heads), dim3(256), 0, st, qs, bias, maskbits, os, lse2, n, heads, n_win, windows, scale);
  else hipLaunchKernelGGL(win_attn_fwd<16>, dim3(per_head, heads), dim3(256), 0, st, qs, bias, maskbits, os, lse2, n, heads, n_win, windows, scale);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_win_attn_backward(const void* qkv, const void* out, const void* dout, const float* lse2, const float* bias,
                                          const unsigned* maskbits, void* dqkv, float* dbias, int windows, int n_win, int n, int heads,
                                          int head_dim, float scale, void* hip_stream) {
  if (!qkv || !out || !dout || !lse2 || !bias || !dqkv || !dbias) return TRANSOAR_ATTN_ERR_NULL;
  if ((head_dim != 16 && head_dim != 32) || n <= 0 || n > kWinN || heads <= 0 || heads > 65535 || windows <= 0 || n_win <= 0 || !(scale > 0.f)) return TRANSOAR_ATTN_ERR_DIM;
  // one resident set of workgroups (80 / 96 KiB of LDS and > 256 registers: one per CU), the windows of a head dealt round-robin
  const int per_head = std::max(1, std::min(windows, 256 / std::min(heads, 256)));
  auto qs = static_cast<const unsigned short*>(qkv);
  auto os = static_cast<const unsigned short*>(out);
  auto ds = static_cast<const unsigned short*>(dout);
  auto dq = static_cast<unsigned short*>(dqkv);
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  // head dimension 16: eight waves per workgroup (0.86 -> 0.77 ms on the stage-0 shape); TRANSOAR_WIN_BWD8=0 keeps four
  static const bool eight = [] { const char* e = getenv("TRANSOAR_WIN_BWD8"); return !(e && e[0] == '0'); }();
  if (head_dim == 32) hipLaunchKernelGGL(win_attn_bwd<32>, dim3(per_head, heads), dim3(256), 0, st, qs, os, ds, lse2, bias, maskbits, dq, dbias, n, heads, n_win, windows, scale);
  else if (eight) hipLaunchKernelGGL(win_attn_bwd8<16>, dim3(per_head, heads), dim3(512), 0, st, qs, os, ds, lse2, bias, maskbits, dq, dbias, n, heads, n_win, windows, scale);
  else hipLaunchKernelGGL(win_attn_bwd<16>, dim3(per_head, heads), dim3(256), 0, st, qs, os, ds, lse2, bias, maskbits, dq, dbias, n, heads, n_win, windows, scale);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_attn_abi_version(void) { return 2; }
