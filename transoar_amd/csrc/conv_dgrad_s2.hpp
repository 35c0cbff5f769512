// Data gradient of a 3x3x3 / stride 2 / pad 1 convolution with FEW input channels (Cin <= 32) and Cout = 16 KS <= 48:
// the 24 -> 48 layer that opens stage 1 of the encoder (backbones/encoder_blocks.py:28-51), whose dx is the largest
// tensor of the backward pass (2 x 160 x 160 x 256 voxels x 24 channels = 629 MB).  HBM bound: per dx voxel 2 Cin
// bytes written + Cout / 4 bytes of dy read; the matrix-core work (27/8 taps x Cout x Cin MACs per voxel) is ~50 us.
//
// The generic path (conv3d_igemm_kernel, classes != 0) runs each parity class of the dx voxels as its own set of
// 128 x 64 tiles: N = Cin = 24 fills 37 % of the tile's MFMAs and every class writes 48-byte rows at a 96-byte stride
// (1.42 ms).  Here ONE workgroup produces all eight classes of a dx tile, so that dx leaves as whole 128-byte lines:
//   tile      2 x 4 x 8 dy voxels (jd, jh, jw) -> 4 x 8 x 16 dx voxels; the dy halo (3 x 5 x 9 voxels, Cout channels,
//             zero beyond the map) is staged once in LDS
//   classes   dx voxel 2 j + p (p = parity per axis): p = 0 sees filter tap 1 of dy voxel j; p = 1 sees tap 2 of dy
//             voxel j and tap 0 of dy voxel j + 1 -> 1, 2, 4 or 8 taps per class, 27 in total
//   waves     4, each owns a fixed set of classes (8 | 4+2+1 | 4+2 | 4+2 taps) and keeps ITS filter slabs as MFMA A
//             fragments in registers for the whole (persistent) kernel: D[channel][voxel] = W[tap]^T (32 x 16 KS) x
//             dy (16 KS x 32 voxels), v_mfma_f32_32x32x16_bf16, one ds_read_b128 of the halo per MFMA
//   epilogue  a lane holds 4 consecutive channels of one voxel per accumulator quad -> 8-byte LDS writes into the dense
//             dx tile, then the tile goes out as 16-byte pieces of contiguous W-runs (16 voxels x 2 Cin bytes)
// Included by conv_gemm.hip (inside its anonymous namespace).
#pragma once

constexpr int kDs2TD = 2, kDs2TH = 4, kDs2TW = 8;                   // dy voxels per tile
constexpr int kDs2HaloRows = (kDs2TD + 1) * (kDs2TH + 1) * (kDs2TW + 1);      // 135
constexpr int kDs2OutVox = 8 * kDs2TD * kDs2TH * kDs2TW;             // 512

__host__ __device__ constexpr int ds2_ntaps(int c) { return c < 0 ? 0 : 1 << (((c >> 2) & 1) + ((c >> 1) & 1) + (c & 1)); }
// e-th tap of class c (bit a of c = parity of axis a: 0 = w, 1 = h, 2 = d): offset of the dy voxel along axis a
__host__ __device__ constexpr int ds2_off(int c, int e, int axis) {
  int bit = 0;
  for (int a = 0; a < axis; ++a) bit += (c >> a) & 1;
  return ((c >> axis) & 1) ? ((e >> bit) & 1) : 0;
}
__host__ __device__ constexpr int ds2_filter(int c, int e, int axis) { return ((c >> axis) & 1) ? (ds2_off(c, e, axis) ? 0 : 2) : 1; }
__host__ __device__ constexpr int ds2_slab(int c, int e) { return (ds2_filter(c, e, 2) * 3 + ds2_filter(c, e, 1)) * 3 + ds2_filter(c, e, 0); }

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned ds2_pack(float lo, float hi) {
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}

// filter slabs of this wave's classes as A fragments: lane (n = lane & 31, kg = lane >> 5) holds wkt[slab][n][16 ks + 8 kg ..+8]
template <int KS, int C0, int C1, int C2>
__device__ __forceinline__ void ds2_load_filter(s16x8 (&a)[8][KS], const unsigned short* __restrict__ wkt, int cin, int lane) {
  constexpr int cls[3] = {C0, C1, C2};
  const int n = lane & 31, kg = lane >> 5;
  int ti = 0;
#pragma unroll
  for (int ci = 0; ci < 3; ++ci) {
#pragma unroll
    for (int e = 0; e < ds2_ntaps(cls[ci]); ++e) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (n < cin) v = *reinterpret_cast<const s16x8*>(wkt + (static_cast<long>(ds2_slab(cls[ci], e)) * cin + n) * (16 * KS) + 16 * ks + 8 * kg);
        a[ti][ks] = v;
      }
      ++ti;
    }
  }
}

template <int KS, int C0, int C1, int C2>
__device__ __forceinline__ void ds2_compute(const s16x8 (&a)[8][KS], const unsigned char* halo, unsigned char* stage, int cin, int lane) {
  constexpr int cls[3] = {C0, C1, C2};
  constexpr int kPitch = 32 * KS + 16;
  const int jw = lane & 7, jh = (lane >> 3) & 3, kg = lane >> 5;
#pragma unroll
  for (int s = 0; s < kDs2TD; ++s) {
    const unsigned char* base = halo + ((s * (kDs2TH + 1) + jh) * (kDs2TW + 1) + jw) * kPitch + kg * 16;
    int ti = 0;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
      if (cls[ci] < 0) continue;
      const int c = cls[ci];
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int e = 0; e < ds2_ntaps(c); ++e) {
        const int off = ((ds2_off(c, e, 2) * (kDs2TH + 1) + ds2_off(c, e, 1)) * (kDs2TW + 1) + ds2_off(c, e, 0)) * kPitch;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = mfma(a[ti][ks], *reinterpret_cast<const s16x8*>(base + off + 32 * ks), acc);
        ++ti;
      }
      // D[row = channel (r & 3) + 8 (r >> 2) + 4 kg][col = this lane's voxel]
      const int zd = 2 * s + ((c >> 2) & 1), zh = 2 * jh + ((c >> 1) & 1), zw = 2 * jw + (c & 1);
      unsigned char* dst = stage + ((zd * (2 * kDs2TH) + zh) * (2 * kDs2TW) + zw) * (cin * 2) + kg * 8;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (8 * g < cin) {
          uint2 v;
          v.x = ds2_pack(acc[4 * g], acc[4 * g + 1]);
          v.y = ds2_pack(acc[4 * g + 2], acc[4 * g + 3]);
          *reinterpret_cast<uint2*>(dst + 16 * g) = v;
        }
    }
  }
}

template <int KS>
__global__ __launch_bounds__(256, 2) void conv3d_dgrad_s2_halo_kernel(
    const unsigned short* __restrict__ DY, const unsigned short* __restrict__ wkt, unsigned short* __restrict__ DX, int N, int OD, int OH,
    int OW, int D, int H, int W, int cin, int tiles_d, int tiles_h, int tiles_w, unsigned dy_bytes) {
  constexpr int kPitch = 32 * KS + 16, kPieces = 2 * KS;         // 16-byte pieces per dy row
  __shared__ __attribute__((aligned(16))) unsigned char halo[kDs2HaloRows * kPitch];
  __shared__ __attribute__((aligned(16))) unsigned char stage[kDs2OutVox * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(DY), 0, static_cast<int>(dy_bytes), 0x00020000);

  s16x8 a[8][KS];
  switch (wave) {
    case 0: ds2_load_filter<KS, 7, -1, -1>(a, wkt, cin, lane); break;
    case 1: ds2_load_filter<KS, 3, 4, 0>(a, wkt, cin, lane); break;
    case 2: ds2_load_filter<KS, 5, 2, -1>(a, wkt, cin, lane); break;
    default: ds2_load_filter<KS, 6, 1, -1>(a, wkt, cin, lane); break;
  }

  const int row_bytes = cin * 2;                                  // dx voxel
  const int run_pieces = 2 * kDs2TW * row_bytes / 16;             // 16-byte pieces of one W-run of the dx tile
  const int out_pieces = 4 * kDs2TD * kDs2TH * run_pieces;
  const long n_tiles = static_cast<long>(N) * tiles_d * tiles_h * tiles_w;
  // the 16-byte pieces of the dx tile this thread stores: the same for every tile (stage offset = 16 p)
  constexpr int kStores = kDs2OutVox * 64 / 16 / 256;             // 8 at Cin = 32
  unsigned out_off[kStores];                                      // < 2^31: checked by the host
  int out_zz[kStores];
#pragma unroll
  for (int k = 0; k < kStores; ++k) {
    const int p = tid + 256 * k;
    const int run = p / run_pieces, pc = p - run * run_pieces;
    const int zd = run / (2 * kDs2TH), zh = run - zd * (2 * kDs2TH);
    out_off[k] = static_cast<unsigned>((zd * H + zh) * W * row_bytes + pc * 16);
    out_zz[k] = p < out_pieces ? (zd | (zh << 8) | ((pc * 16 / row_bytes) << 16)) : -1;        // -1: no piece (Cin < 32)
  }
  constexpr int kLoads = (kDs2HaloRows * kPieces + 255) / 256;
  u32x4 pre[kLoads];
  int jd0 = 0, jh0 = 0, jw0 = 0, b = 0;
  // dy halo of tile t -> registers (rows beyond the map read as zeros); issued one tile ahead of its use
  auto prefetch = [&](long t) {
    const int tw = static_cast<int>(t % tiles_w);
    long r = t / tiles_w;
    const int th = static_cast<int>(r % tiles_h);
    r /= tiles_h;
    const int td = static_cast<int>(r % tiles_d), bb = static_cast<int>(r / tiles_d);
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      const int p = tid + 256 * k;
      const int row = p / kPieces, pc = p - row * kPieces;
      const int hd = row / ((kDs2TH + 1) * (kDs2TW + 1)), rem = row - hd * ((kDs2TH + 1) * (kDs2TW + 1));
      const int hh = rem / (kDs2TW + 1), hw = rem - hh * (kDs2TW + 1);
      const int jd = td * kDs2TD + hd, jh = th * kDs2TH + hh, jw = tw * kDs2TW + hw;
      const bool ok = p < kDs2HaloRows * kPieces && jd < OD && jh < OH && jw < OW;
      const unsigned off = static_cast<unsigned>(((bb * OD + jd) * OH + jh) * OW + jw) * static_cast<unsigned>(32 * KS) + pc * 16;
      pre[k] = __builtin_amdgcn_raw_buffer_load_b128(rdy, ok ? off : 0x80000000u, 0, 0);
    }
  };
  if (blockIdx.x < n_tiles) prefetch(blockIdx.x);
  for (long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    {
      const int tw = static_cast<int>(t % tiles_w);
      long r = t / tiles_w;
      const int th = static_cast<int>(r % tiles_h);
      r /= tiles_h;
      jd0 = static_cast<int>(r % tiles_d) * kDs2TD; jh0 = th * kDs2TH; jw0 = tw * kDs2TW; b = static_cast<int>(r / tiles_d);
    }
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      const int p = tid + 256 * k;
      const int row = p / kPieces, pc = p - row * kPieces;
      if (p < kDs2HaloRows * kPieces) *reinterpret_cast<u32x4*>(halo + row * kPitch + pc * 16) = pre[k];
    }
    __syncthreads();
    if (t + gridDim.x < n_tiles) prefetch(t + gridDim.x);       // in flight during the MFMA phase
    switch (wave) {
      case 0: ds2_compute<KS, 7, -1, -1>(a, halo, stage, cin, lane); break;
      case 1: ds2_compute<KS, 3, 4, 0>(a, halo, stage, cin, lane); break;
      case 2: ds2_compute<KS, 5, 2, -1>(a, halo, stage, cin, lane); break;
      default: ds2_compute<KS, 6, 1, -1>(a, halo, stage, cin, lane); break;
    }
    __syncthreads();
    // ---- dense dx tile -> global: W-runs of 16 voxels (2 Cin x 16 bytes, contiguous)
    {
      const int xd0 = 2 * jd0, xh0 = 2 * jh0, xw0 = 2 * jw0;
      unsigned char* tile = reinterpret_cast<unsigned char*>(DX) + (((static_cast<long>(b) * D + xd0) * H + xh0) * W + xw0) * row_bytes;
#pragma unroll
      for (int k = 0; k < kStores; ++k) {
        const int zd = out_zz[k] & 0xff, zh = (out_zz[k] >> 8) & 0xff, vox = out_zz[k] >> 16;
        if (out_zz[k] >= 0 && xd0 + zd < D && xh0 + zh < H && xw0 + vox < W)
          *reinterpret_cast<u32x4*>(tile + out_off[k]) = *reinterpret_cast<const u32x4*>(stage + (tid + 256 * k) * 16);
      }
    }
    // no barrier here: the next tile's halo stores touch only `halo` (last read before the barrier above), and `stage`
    // is rewritten only after the next tile's first barrier, which every thread reaches after its reads above
  }
}
