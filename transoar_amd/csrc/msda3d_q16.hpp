// Forward gather of MSDeformAttn-3D on the matrix cores, point-column form with 16 queries per wave (gfx950, round 6).
//
// Semantics: SURVEY.md appendix A (ops/src/cuda/ms_deform_im2col_cuda.cuh:31-114, 370-439).
//
// MEASURED SLOWER than msda3d_pcm.hpp (1.05 / 0.81 ms against 0.66 / 0.50 at the flagship shape) and
// therefore behind TRANSOAR_MSDA3D_Q16: this is the speed-of-light probe of the carving round 5's review asked for, kept
// parity-tested.  What it set out to do against msda3d_pcm.hpp (8 queries per wave, 1 363 vector instructions and 256 staged
// rows per wave):
//
//   * one wave owns 16 queries (a 2x2x4 sub-brick of a 4x4x8 brick) of one head and walks UPW consecutive units (the sub-bricks
//     of one brick first).  The box of 16 queries turned out 1.29x the box of 8 (322 against 250 rows on jittered locations):
//     rows staged per query fall 1.55x, matrix work per query rises 1.27x;
//   * lane = (query, point) = one MFMA column, for all four levels: 4 points of geometry per lane, no exchange
//     between wave halves, its 8 corner weights per level go to its OWN weight column in LDS;
//   * the weight block is two bf16 planes (hi and lo halves of every weight, 2^-16 together) [column][K-slot]:
//     the MFMA A fragment of a lane is one ds_read_b128 per plane and group -- no v_perm to separate the halves;
//   * the 64 columns are two MFMA groups (d-layer 0 / 1 of the sub-brick) over the same staged rows: 8 MFMAs
//     32x32x16 per 16 rows (2 groups x 2 channel halves x hi/lo);
//   * the unit's loc / attn words come straight into registers (4 x 12 + 4 x 4 bytes per lane), and the NEXT
//     unit's are requested before the current unit's rows: a wave's first memory round trip is off its path;
//   * unit decode is scalar: the integer divisions are multiplications by host-made reciprocals.
//
// As in the round-3 kernel: per level the box of corner voxels (out-of-level voxels staged as zero rows) is walked
// in K-blocks of 32 rows, global -> registers -> LDS as whole 128-byte head slices, prefetched a block ahead and
// across levels; B = V through the transposing ds_read_b64_tr_b16; a level whose box exceeds kQ16BoxRows rows runs
// the same loop over an explicit (column, corner) row list; zero weights never widen the box (integer pixel
// coordinates: the module's initial state, ms_deform_attn.py:67-82).
#pragma once
#include "msda3d_common.hpp"
#include "msda3d_mma.hpp"
#include "msda3d_pcm.hpp"

namespace transoar {

constexpr int kQ16KB = 32;                // value rows per K-block = K-slots of a weight column
constexpr int kQ16WS = 80;                // bytes per weight column and plane: 32 bf16 K-slots + 16 bytes of padding (the 16-byte reads of 16 consecutive
                                          // columns tile the 64 banks).  The padding is also the guard of the pair stores: K-slot 32 of a column is its
                                          // first padding slot, K-slot -1 the last padding slot of the column before it (of the 16 bytes in front of column 0)
constexpr int kQ16W0 = 16;                // byte offset of column 0 inside a plane
constexpr int kQ16Plane = 64 * kQ16WS + 16;    // bytes per plane (64 columns)
constexpr int kQ16BoxRows = 256;          // larger boxes: explicit (column, corner) slots, 512 per level
constexpr int kQ16Far = 0x20000000;       // K-slot of a skipped point: in no block
constexpr unsigned kQ16Oob = 0xffffff00u; // byte offset past every buffer, still past it with a small immediate added

// Launch constants (device memory, scalar loads)
struct Q16Const {
  PcmConst pc;
  unsigned long long mg_M, mg_bricks, mg_nbw[4], mg_nbh[4];     // floor(2^40 / d) + 1: u / d = (u * mg) >> 40 for u < 2^24
};
__device__ __forceinline__ unsigned q16_div(unsigned u, unsigned long long mg) {
  return static_cast<unsigned>((static_cast<unsigned long long>(u) * mg) >> 40);
}

struct Q16Unit {          // wave-uniform: one work unit = a 2x2x4 sub-brick of queries x one head
  int lq, od, oh, ow, qD, qH, qW, qbase;
  unsigned b, m;
  bool any;               // false: the whole sub-brick is padding
};

// PROBE (measurement only): 0 = the kernel; 1 = no parameter stream (locations made from the query's own position);
// 2 = no geometry either (synthetic boxes, slots and weights): staging + matrix work + stores alone.
template <typename VT, int PROBE>
__global__ __launch_bounds__(64, 2) void msda3d_fwd_q16(
    const VT* __restrict__ value, const float* __restrict__ loc, const float* __restrict__ attn, VT* __restrict__ out,
    int S, int M, int L, unsigned value_bytes, unsigned loc_bytes, unsigned attn_bytes, unsigned n_units, unsigned upw,
    const Q16Const* __restrict__ cst) {
  const BrickOrder& order = cst->pc.order;
  constexpr int C = 64, KB = kQ16KB, VP = 128, WS = kQ16WS;
  __shared__ __attribute__((aligned(128))) unsigned char vbuf[KB * VP];           // staged rows (swizzled as in msda3d_pcm.hpp); the output rows alias it
  __shared__ __attribute__((aligned(16))) unsigned char wbuf[2 * kQ16Plane];      // [plane hi | lo][column][K-slot] bf16

  // XCD-contiguous wave order (block b runs on XCD b % 8: each XCD walks one contiguous eighth)
  const unsigned n_waves = (n_units + upw - 1u) / upw;
  const unsigned per_xcd = (n_waves + 7u) >> 3;
  const unsigned wv = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (wv >= n_waves) return;
  unsigned u = wv * upw;
  const unsigned u_end = min(u + upw, n_units);

  const int lane = threadIdx.x;
  const int kh = lane >> 5, n = lane & 31, q = lane >> 2, p = lane & 3;
  const int dq = q >> 3, hq = (q >> 2) & 1, wq = q & 3;                  // the query's place in the 2x2x4 sub-brick
  const unsigned bricks = static_cast<unsigned>(order.pad_start[order.L]) >> 7;
  const unsigned row_bytes = static_cast<unsigned>(M) * C * sizeof(VT);
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<VT*>(value), 0, static_cast<int>(value_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(loc), 0, static_cast<int>(loc_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(attn), 0, static_cast<int>(attn_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(out, 0, static_cast<int>(value_bytes), 0x00020000);

  auto decode = [&](unsigned uu) -> Q16Unit {
    Q16Unit un;
    const unsigned sub = uu & 7u, t1 = uu >> 3;
    const unsigned t2 = q16_div(t1, cst->mg_M);
    un.m = t1 - t2 * static_cast<unsigned>(M);
    un.b = q16_div(t2, cst->mg_bricks);
    const int brick = static_cast<int>(bricks - 1u - (t2 - un.b * bricks));     // coarse levels first: their boxes are the big ones
    int lq = 0;
#pragma unroll
    for (int t = 1; t < kPcmLevels; ++t) lq += (t < order.L && brick * kBrickSlots >= order.pad_start[t]) ? 1 : 0;
    un.lq = lq;
    un.qD = order.D[lq]; un.qH = order.H[lq]; un.qW = order.W[lq];
    const unsigned lbrick = static_cast<unsigned>(brick - (order.pad_start[lq] >> 7));
    const unsigned brest = q16_div(lbrick, cst->mg_nbw[lq]);
    const unsigned bwi = lbrick - brest * static_cast<unsigned>(order.nbw[lq]);
    const unsigned bdi = q16_div(brest, cst->mg_nbh[lq]);
    const unsigned bhi = brest - bdi * static_cast<unsigned>(order.nbh[lq]);
    un.od = static_cast<int>(bdi * kBrickD + 2 * (sub >> 2));
    un.oh = static_cast<int>(bhi * kBrickH + 2 * ((sub >> 1) & 1));
    un.ow = static_cast<int>(bwi * kBrickW + 4 * (sub & 1));
    un.qbase = order.start[lq] + static_cast<int>(un.b) * S;
    un.any = un.od < un.qD && un.oh < un.qH && un.ow < un.qW;
    return un;
  };
  // b * S + pyramid row of query (dd, hh, ww) of the sub-brick, -1 = padding
  auto query_row = [&](const Q16Unit& un, int dd, int hh, int ww) -> int {
    const int d = un.od + dd, h = un.oh + hh, w = un.ow + ww;
    const int r = un.qbase + __mul24(__mul24(d, un.qH) + h, un.qW) + w;
    return (d < un.qD && h < un.qH && w < un.qW) ? r : -1;
  };
  typedef float f32x3_t __attribute__((ext_vector_type(3)));
  struct Params {
    f32x3_t lc[kPcmLevels];
    float at[kPcmLevels];
  };
  // the lane's 4 locations and 4 weights of unit `un` (point p of query q on every level), non-temporal: they pass through once
  auto issue_params = [&](const Q16Unit& un, int s) -> Params {
    Params pr;
    const unsigned item = __umul24(static_cast<unsigned>(s), static_cast<unsigned>(M)) + un.m;
    const bool ok = s >= 0 && un.any;
    const unsigned loff = ok ? item * (static_cast<unsigned>(L) * 48u) + static_cast<unsigned>(p) * 12u : kQ16Oob;
    const unsigned aoff = ok ? item * (static_cast<unsigned>(L) * 16u) + static_cast<unsigned>(p) * 4u : kQ16Oob;
#pragma unroll
    for (int l = 0; l < kPcmLevels; ++l) {
      // a level >= L reads the next item's words (or past the end: zeros): it is never used
      pr.lc[l] = __builtin_bit_cast(f32x3_t, __builtin_amdgcn_raw_buffer_load_b96(lrs, loff + static_cast<unsigned>(l) * 48u, 0, 2));
      pr.at[l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ars, aoff + static_cast<unsigned>(l) * 16u, 0, 2));
    }
    return pr;
  };

  // ---- once per wave: clear the weight planes
  for (int i = lane; i < 2 * kQ16Plane / 16; i += 64) reinterpret_cast<u32x4*>(wbuf)[i] = u32x4{0u, 0u, 0u, 0u};

  // addressing of the row staging (as msda3d_pcm.hpp): 8 lanes fetch one 128-byte row, 16 bytes each
  const int st_row = lane >> 3, st_vec = lane & 7;
  const int st_swz = (st_vec ^ ((st_row & 2) << 1)) * 16;
  const int tr_off0 = ((16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2) ^ ((((lane & 15) >> 2) & 2) << 5), tr_off1 = tr_off0 ^ 64;
  unsigned char* const wcol = wbuf + kQ16W0 + lane * WS - 2;                      // K-slot -1 of this lane's weight column (hi plane)
  const unsigned char* const wrd = wbuf + kQ16W0 + n * WS + kh * 16;              // its A-fragment reads: column n (+32), K-slots 8 kh ..

  Q16Unit un = decode(u);
  int s = query_row(un, dq, hq, wq);
  Params pr;
  if constexpr (PROBE == 0) pr = issue_params(un, s);

  for (;;) {
    const bool live = s >= 0 && un.any;
    const unsigned head_off = (un.b * static_cast<unsigned>(S) * M + un.m) * (C * sizeof(VT));

    // ---- geometry of this lane's four points: (level l, point p) of query q
    float fl[kPcmLevels][3], fa[kPcmLevels];     // fractional parts (d, h, w), attention weight
    int dhw[kPcmLevels];                         // (d0 + 1) | (h0 + 1) << 10 | (w0 + 1) << 20, kPcmSkip for a skipped point
    PcmBox box[kPcmLevels];
    int mode[kPcmLevels];                        // 0 = no valid point, 1 = box rows, 2 = explicit (column, corner) slots
    if constexpr (PROBE < 2) {
      int i0[kPcmLevels][3], t0[kPcmLevels][3];
      bool okp[kPcmLevels];
#pragma unroll
      for (int l = 0; l < kPcmLevels; ++l) {
        const float fW = cst->pc.fW[l], fH = cst->pc.fH[l], fD = cst->pc.fD[l];
        float px, py, pz;
        if constexpr (PROBE == 0) {
          px = pr.lc[l][0]; py = pr.lc[l][1]; pz = pr.lc[l][2];
          fa[l] = pr.at[l];
        } else {                                 // the refine block's pattern: voxel centre + (p + 1) voxels along the head's axis + 0.3
          const float cx = (static_cast<float>(un.ow + wq) + 0.5f) / static_cast<float>(un.qW);
          const float cy = (static_cast<float>(un.oh + hq) + 0.5f) / static_cast<float>(un.qH);
          const float cz = (static_cast<float>(un.od + dq) + 0.5f) / static_cast<float>(un.qD);
          const float st = static_cast<float>(p + 1) * ((un.m & 1u) ? 1.f : -1.f);
          px = cx + ((un.m >> 1) == 0 ? st + 0.3f : 0.3f) / fW;
          py = cy + ((un.m >> 1) == 1 ? st + 0.3f : 0.3f) / fH;
          pz = cz + ((un.m >> 1) == 2 ? st + 0.3f : 0.3f) / fD;
          fa[l] = 0.0625f;
        }
        const float w_im = pixel_coord_f(px, fW), h_im = pixel_coord_f(py, fH), d_im = pixel_coord_f(pz, fD);
        const bool ok = live && l < L && d_im > -1.f && h_im > -1.f && w_im > -1.f && d_im < fD && h_im < fH && w_im < fW;
        const float fd = floorf(d_im), fh = floorf(h_im), fw = floorf(w_im);
        i0[l][0] = static_cast<int>(fd); i0[l][1] = static_cast<int>(fh); i0[l][2] = static_cast<int>(fw);
        fl[l][0] = d_im - fd; fl[l][1] = h_im - fh; fl[l][2] = w_im - fw;
        fa[l] = ok ? fa[l] : 0.f;
        okp[l] = ok;
        dhw[l] = ok ? (i0[l][0] + 1) | ((i0[l][1] + 1) << 10) | ((i0[l][2] + 1) << 20) : kPcmSkip;
        // highest corner voxel the point really touches along an axis: i0 + 1 only if its fraction is non-zero
#pragma unroll
        for (int k = 0; k < 3; ++k) t0[l][k] = i0[l][k] + (fl[l][k] > 0.f ? 1 : 0);
      }
      // ---- boxes: wave-wide minima of the packed (level 2j, level 2j + 1) corner coordinates and their negatives
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned okmask = (okp[2 * j] ? 0x0000ffffu : 0u) | (okp[2 * j + 1] ? 0xffff0000u : 0u);
        int lo3[2][3], hi3[2][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const unsigned pk = __builtin_amdgcn_perm(static_cast<unsigned>(i0[2 * j + 1][k]), static_cast<unsigned>(i0[2 * j][k]), 0x05040100u);
          const unsigned tp = __builtin_amdgcn_perm(static_cast<unsigned>(t0[2 * j + 1][k]), static_cast<unsigned>(t0[2 * j][k]), 0x05040100u);
          const unsigned ng = __builtin_bit_cast(unsigned, -__builtin_bit_cast(s16x2, tp));
          const int a = __builtin_amdgcn_readlane(wave_min_pk16(static_cast<int>((pk & okmask) | (0x7fff7fffu & ~okmask))), 63);
          const int c = __builtin_amdgcn_readlane(wave_min_pk16(static_cast<int>((ng & okmask) | (0x7fff7fffu & ~okmask))), 63);
          lo3[0][k] = static_cast<short>(a); lo3[1][k] = a >> 16;
          hi3[0][k] = -static_cast<int>(static_cast<short>(c)); hi3[1][k] = -(c >> 16);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int l = 2 * j + i;
          box[l] = PcmBox{lo3[i][0], lo3[i][1], lo3[i][2], hi3[i][0] - lo3[i][0] + 1, hi3[i][1] - lo3[i][1] + 1, hi3[i][2] - lo3[i][2] + 1};
          mode[l] = (l >= L || lo3[i][0] == 32767) ? 0 : (box[l].TD * box[l].TH * box[l].TW <= kQ16BoxRows ? 1 : 2);
        }
      }
    } else {
      // PROBE 2: boxes from the sub-brick's origin alone, slots from the lane index, constant weights
#pragma unroll
      for (int l = 0; l < kPcmLevels; ++l) {
        const int D = order.D[l], H = order.H[l], W = order.W[l];
        const int bd = un.od * D / un.qD - 1, bh = un.oh * H / un.qH - 1, bw = un.ow * W / un.qW - 1;
        const bool same = l == un.lq;
        box[l] = PcmBox{bd, bh, bw, 3, 3, same ? 9 : (l > un.lq ? 7 : 8)};
        mode[l] = (l < L && un.any) ? 1 : 0;
        fl[l][0] = fl[l][1] = fl[l][2] = 0.5f;
        fa[l] = 0.0625f;
        dhw[l] = (bd + dq + 1) | ((bh + hq + 1) << 10) | ((bw + (same ? wq : (wq >> 1)) + p + 1) << 20);
      }
    }

    // ---- the next unit's parameters: requested now, used after this unit's rows
    const unsigned u_nx = u + 1u;
    const bool more = u_nx < u_end;

    f32x16 acc[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[g][hh][i] = 0.f;

    // Byte offsets of the 64 rows [k0, k0 + 64) of level l (lane i: row k0 + i); a row past the end or outside the
    // level gets an offset past the buffer that stays past it when the 16-byte piece offset is added.
    auto row_offsets = [&](auto lc, int k0) -> int {
      constexpr int l = decltype(lc)::value;
      const PcmBox bx = box[l];
      const int D = order.D[l], H = order.H[l], W = order.W[l], start = order.start[l];
      const int r = k0 + lane;
      int d, h, w;
      bool ok;
      if (mode[l] == 1) {
        const int THW = bx.TH * bx.TW, R = bx.TD * THW;
        // r -> (rd, rh, rw) by float reciprocals: (r + 0.5) / n is >= 0.5 / n away from an integer, far more than
        // the float error for r < 2^12
        const float inv_thw = __builtin_amdgcn_rcpf(static_cast<float>(THW)), inv_tw = __builtin_amdgcn_rcpf(static_cast<float>(bx.TW));
        const int rd = static_cast<int>((static_cast<float>(r) + 0.5f) * inv_thw), rr = r - __mul24(rd, THW);
        const int rh = static_cast<int>((static_cast<float>(rr) + 0.5f) * inv_tw), rw = rr - __mul24(rh, bx.TW);
        d = bx.bd + rd; h = bx.bh + rh; w = bx.bw + rw;
        ok = r < R;
      } else {
        // explicit slots: slot r = column (r >> 3) = lane (r >> 3), corner (r & 7) = dd*4 + dh*2 + dw
        const int pd = __builtin_amdgcn_ds_bpermute((r >> 3) * 4, dhw[l]);
        d = (pd & 1023) - 1 + ((r >> 2) & 1); h = ((pd >> 10) & 1023) - 1 + ((r >> 1) & 1); w = ((pd >> 20) & 1023) - 1 + (r & 1);
        ok = pd != kPcmSkip;
      }
      ok = ok && static_cast<unsigned>(d) < static_cast<unsigned>(D) && static_cast<unsigned>(h) < static_cast<unsigned>(H) &&
           static_cast<unsigned>(w) < static_cast<unsigned>(W);
      const int grow = start + __mul24(__mul24(d, H) + h, W) + w;
      return ok ? static_cast<int>(head_off + __umul24(static_cast<unsigned>(grow), row_bytes)) : static_cast<int>(kQ16Oob);
    };
    auto issue_loads = [&](int row_off, int half, u32x4 (&pre)[4]) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const unsigned off = static_cast<unsigned>(__builtin_amdgcn_ds_bpermute((half * 32 + it * 8 + st_row) * 4, row_off)) + st_vec * 16u;
        pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
      }
    };

    u32x4 pre[4];
    bool have_pre = false;
    int row_off = 0;
    static_for<0, kPcmLevels>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      // the next unit's parameters are requested when the last level begins: by then the registers of the earlier
      // levels' geometry are free, and the words have this level's blocks and the output pass to arrive in
      if (l + 1 == L && more) {
        if constexpr (PROBE == 0) {
          const Q16Unit un_nx = decode(u_nx);
          pr = issue_params(un_nx, query_row(un_nx, dq, hq, wq));
        }
      }
      if (mode[l] == 0) return;
      const PcmBox bx = box[l];
      const int THW = bx.TH * bx.TW;
      const int R = mode[l] == 1 ? bx.TD * THW : 512;
      // ---- the lane's eight entries on this level: four (dw = 0, dw = 1) pairs in adjacent K-slots.  c1 = first
      // K-slot + 1, sp[j] the (wave-uniform) slot offset of pair j = 2 dd + dh.  A pair is written when either slot is in the
      // block; the other one then lands in a guard slot of the column (K-slot -1 or 32), which is never read.
      // A corner whose fraction is zero does not widen the box (t0 above); its slot may lie outside the box and alias
      // another slot of the column.  Such a ghost never destroys a real weight: slot offsets grow with the corner
      // number (TH*TW >= TW >= 1) and the entries are written in corner order, so a real entry that shares a ghost's slot
      // is always written after it.
      int c1, sp[4];
      if (mode[l] == 1) {
        const int pd = dhw[l];
        const int d1 = pd & 1023, h1 = (pd >> 10) & 1023, w1 = pd >> 20;        // d0 + 1, h0 + 1, w0 + 1
        const int base = __mul24(__mul24(bx.bd + 1, bx.TH) + (bx.bh + 1), bx.TW) + (bx.bw + 1) - 1;
        c1 = pd == kPcmSkip ? kQ16Far : __mul24(__mul24(d1, bx.TH) + h1, bx.TW) + w1 - base;
        sp[0] = 0; sp[1] = bx.TW; sp[2] = THW; sp[3] = THW + bx.TW;
      } else {
        c1 = dhw[l] == kPcmSkip ? kQ16Far : lane * 8 + 1;
        sp[0] = 0; sp[1] = 2; sp[2] = 4; sp[3] = 6;
      }
      unsigned wq8[8];
      {
        const float ld = fl[l][0], lh = fl[l][1], lw = fl[l][2];
        const float wd[2] = {fa[l] * (1.f - ld), fa[l] * ld}, wh[2] = {1.f - lh, lh}, ww[2] = {1.f - lw, lw};
#pragma unroll
        for (int e = 0; e < 8; ++e) wq8[e] = PcmW<VT>::split((wd[e >> 2] * wh[(e >> 1) & 1]) * ww[e & 1]);
      }
      const int nb = (R + KB - 1) / KB;
      if (!have_pre) {
        row_off = row_offsets(lc, 0);
        issue_loads(row_off, 0, pre);
      }
      for (int hb = 0; hb < nb; ++hb) {
        const int half = hb & 1;                                            // which half of the 64 row offsets in row_off
        const int k0 = hb * KB;
        const int nch = PROBE == 3 ? 2 : (min(KB, R - k0) + 15) >> 4;        // 16-row chunks of this block: 1 or 2
        // ---- staged rows -> LDS
#pragma unroll
        for (int it = 0; it < 2; ++it) *reinterpret_cast<u32x4*>(vbuf + (it * 8 + st_row) * VP + st_swz) = pre[it];
        if (nch > 1) {
#pragma unroll
          for (int it = 2; it < 4; ++it) *reinterpret_cast<u32x4*>(vbuf + (it * 8 + st_row) * VP + st_swz) = pre[it];
        }
        // ---- the lane's pairs that touch this block: t1 = (K-slot of the pair's first entry) - k0 + 1 in [0, 32]
        int t1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          t1[j] = c1 + (sp[j] - k0);
          if (PROBE != 3 && static_cast<unsigned>(t1[j]) < static_cast<unsigned>(KB + 1)) {
            // (two 16-bit stores per plane on purpose: hipcc would merge adjacent ones into a 32-bit store at a 2-byte
            // aligned address, which the LDS executes -- correctly -- at less than half the rate)
            unsigned char* a = wcol + 2 * t1[j];
            unsigned char* a2 = a;
            asm volatile("" : "+v"(a2));
            *reinterpret_cast<unsigned short*>(a) = static_cast<unsigned short>(wq8[2 * j] >> 16);
            *reinterpret_cast<unsigned short*>(a2 + 2) = static_cast<unsigned short>(wq8[2 * j + 1] >> 16);
            *reinterpret_cast<unsigned short*>(a + kQ16Plane) = static_cast<unsigned short>(wq8[2 * j]);
            *reinterpret_cast<unsigned short*>(a2 + kQ16Plane + 2) = static_cast<unsigned short>(wq8[2 * j + 1]);
          }
        }
        // ---- prefetch the next block (of this level, or the first of the next one)
        have_pre = false;
        if (hb + 1 < nb) {
          if (half == 1) row_off = row_offsets(lc, (hb + 1) * KB);
          issue_loads(row_off, half ^ 1, pre);
          have_pre = true;
        } else if constexpr (l + 1 < kPcmLevels) {
          if (mode[l + 1] != 0) {
            row_off = row_offsets(IntC<l + 1>{}, 0);
            issue_loads(row_off, 0, pre);
            have_pre = true;
          }
        }
        // ---- 16-row chunks on the matrix cores (B = V: lane supplies row (lane & 15) >> 2 of its group's 4-row set, 4 channels;
        // receives its channel's column).  The second chunk's fragment reads follow the first chunk's MFMAs into the queue.
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const unsigned char* vrow = vbuf + (8 * kh + ((lane & 15) >> 2)) * VP;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
          if (kc < nch) {
            const unsigned char* vr = vrow + kc * 16 * VP;
            const s16x4 b00 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vr + tr_off0));
            const s16x4 b01 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vr + 4 * VP + tr_off0));
            const s16x4 b10 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vr + tr_off1));
            const s16x4 b11 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vr + 4 * VP + tr_off1));
            const s16x8 v0 = __builtin_shufflevector(b00, b01, 0, 1, 2, 3, 4, 5, 6, 7);
            const s16x8 v1 = __builtin_shufflevector(b10, b11, 0, 1, 2, 3, 4, 5, 6, 7);
            s16x8 whi[2], wlo[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const unsigned char* wp = wrd + g * 32 * WS + kc * 32;
              whi[g] = __builtin_bit_cast(s16x8, *reinterpret_cast<const u32x4*>(wp));
              wlo[g] = __builtin_bit_cast(s16x8, *reinterpret_cast<const u32x4*>(wp + kQ16Plane));
            }
            acc[0][0] = Mma<VT>::mfma(whi[0], v0, acc[0][0]);
            acc[0][1] = Mma<VT>::mfma(whi[0], v1, acc[0][1]);
            acc[1][0] = Mma<VT>::mfma(whi[1], v0, acc[1][0]);
            acc[1][1] = Mma<VT>::mfma(whi[1], v1, acc[1][1]);
            acc[0][0] = Mma<VT>::mfma(wlo[0], v0, acc[0][0]);
            acc[0][1] = Mma<VT>::mfma(wlo[0], v1, acc[0][1]);
            acc[1][0] = Mma<VT>::mfma(wlo[1], v0, acc[1][0]);
            acc[1][1] = Mma<VT>::mfma(wlo[1], v1, acc[1][1]);
          }
        }
        // ---- clear the block's pairs again (the test is made again from t1: lane masks kept across the matrix work
        // would cost scalar registers the kernel does not have)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int t = t1[j];
          asm volatile("" : "+v"(t));
          if (PROBE != 3 && static_cast<unsigned>(t) < static_cast<unsigned>(KB + 1)) {
            unsigned char* a = wcol + 2 * t;
            unsigned char* a2 = a;
            asm volatile("" : "+v"(a2));
            *reinterpret_cast<unsigned short*>(a) = 0;
            *reinterpret_cast<unsigned short*>(a2 + 2) = 0;
            *reinterpret_cast<unsigned short*>(a + kQ16Plane) = 0;
            *reinterpret_cast<unsigned short*>(a2 + kQ16Plane + 2) = 0;
          }
        }
      }
    });

    // ---- D[(query, point)][channel]: register r of a lane is column (r & 3) + 8 (r >> 2) + 4 kh of its group = query
    // 2 (r >> 2) + kh of the group, point r & 3, channel lane & 31 (+ 32 for the second tile).  Sum the points, rows
    // leave through LDS as whole 128-byte lines.
    if (un.any) {
      unsigned short* ob = reinterpret_cast<unsigned short*>(vbuf);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float v0 = (acc[g][0][4 * r4] + acc[g][0][4 * r4 + 1]) + (acc[g][0][4 * r4 + 2] + acc[g][0][4 * r4 + 3]);
          const float v1 = (acc[g][1][4 * r4] + acc[g][1][4 * r4 + 1]) + (acc[g][1][4 * r4 + 2] + acc[g][1][4 * r4 + 3]);
          const unsigned pk = PcmW<VT>::pack2(v0, v1);
          const int qq = g * 8 + 2 * r4 + kh;
          ob[qq * C + n] = static_cast<unsigned short>(pk);
          ob[qq * C + 32 + n] = static_cast<unsigned short>(pk >> 16);
        }
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int qq = it * 8 + (lane >> 3);
        const int sq = query_row(un, qq >> 3, (qq >> 2) & 1, qq & 3);
        const u32x4 line = *reinterpret_cast<const u32x4*>(vbuf + qq * 128 + (lane & 7) * 16);
        const unsigned ooff = sq >= 0 ? (__umul24(static_cast<unsigned>(sq), static_cast<unsigned>(M)) + un.m) * (C * sizeof(VT)) + (lane & 7) * 16u : kQ16Oob;
        __builtin_amdgcn_raw_buffer_store_b128(line, ors, ooff, 0, 2);
      }
    }
    if (!more) break;
    u = u_nx;
    un = decode(u);               // scalar, cheap: made again instead of carried across the levels
    s = query_row(un, dq, hq, wq);
  }
}

}  // namespace transoar
