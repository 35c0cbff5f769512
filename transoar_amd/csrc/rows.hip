// Row gather / inverse-gather for the Focused Decoder's per-organ key lists
// (transoar_amd/focused_decoder.py): out[b][k] = src[b][index[k]] and its
// adjoint grad_src[b][s] = sum_{k : index[k] == s} g[b][k], the latter as a pull
// over a static CSR inverse of `index` (no atomics; each organ's box is fixed at
// model construction).  Rows are C elements of fp32 or bf16, C*elt % 16 == 0;
// one thread moves 16 bytes.  torch's index_select / index_add_ run these at
// 0.2-0.7 TB/s on this shape; these are plain HBM-speed copies.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/transoar_rows.h"

namespace transoar {

using u32x4r = __attribute__((ext_vector_type(4))) unsigned int;

__global__ __launch_bounds__(256) void rows_gather(const u32x4r* __restrict__ src, const int* __restrict__ index,
                                                   u32x4r* __restrict__ out, long S, long K, int vec_per_row) {
  const long t = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  const long k = t / vec_per_row;
  const int v = static_cast<int>(t - k * vec_per_row);
  if (k >= K) return;
  const long b = blockIdx.y;
  const int s = index[k];                       // negative: a zero row (the padding slots of the Swin window layout)
  out[(b * K + k) * vec_per_row + v] = s < 0 ? u32x4r{0u, 0u, 0u, 0u} : src[(b * S + s) * vec_per_row + v];
}

// out[b][k] = resid[b][k] + scale[b] * src[b][index[k]] on bf16 rows (fp32 arithmetic, one rounding): the Swin block's window
// merge with its residual add and the per-sample stochastic-depth factor in the gather's own pass (resid / scale may be null:
// the adjoint is the same gather through the inverse list, scaled, without a residual).
__global__ __launch_bounds__(256) void rows_gather_axpy(const u32x4r* __restrict__ src, const int* __restrict__ index,
                                                        const float* __restrict__ scale, const u32x4r* __restrict__ resid,
                                                        u32x4r* __restrict__ out, long S, long K, int vec_per_row) {
  const long t = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  const long k = t / vec_per_row;
  const int v = static_cast<int>(t - k * vec_per_row);
  if (k >= K) return;
  const long b = blockIdx.y;
  const int s = index[k];
  const long at = (b * K + k) * vec_per_row + v;
  u32x4r x{0u, 0u, 0u, 0u};
  if (s >= 0) x = src[(b * S + s) * vec_per_row + v];
  const float f = scale != nullptr ? scale[b] : 1.f;
  if (resid == nullptr && scale == nullptr) { out[at] = x; return; }
  u32x4r r{0u, 0u, 0u, 0u};
  if (resid != nullptr) r = resid[at];
  u32x4r o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float lo = __builtin_fmaf(__uint_as_float(x[e] << 16), f, __uint_as_float(r[e] << 16));
    const float hi = __builtin_fmaf(__uint_as_float(x[e] & 0xffff0000u), f, __uint_as_float(r[e] & 0xffff0000u));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    o[e] = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
  }
  out[at] = o;
}

template <bool BF16>
__global__ __launch_bounds__(256) void rows_pull_sum(const u32x4r* __restrict__ g, const int* __restrict__ inv_ptr,
                                                     const int* __restrict__ inv_idx, u32x4r* __restrict__ out,
                                                     long S, long K, int vec_per_row) {
  const long t = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  const long s = t / vec_per_row;
  const int v = static_cast<int>(t - s * vec_per_row);
  if (s >= S) return;
  const long b = blockIdx.y;
  const int beg = inv_ptr[s], end = inv_ptr[s + 1];
  constexpr int NE = BF16 ? 8 : 4;
  float acc[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) acc[e] = 0.f;
  for (int i = beg; i < end; ++i) {
    const u32x4r r = g[(b * K + inv_idx[i]) * vec_per_row + v];
    if (BF16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[2 * e] += __uint_as_float(r[e] << 16);
        acc[2 * e + 1] += __uint_as_float(r[e] & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += __uint_as_float(r[e]);
    }
  }
  u32x4r o;
  if (BF16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned lo = __float_as_uint(acc[2 * e]), hi = __float_as_uint(acc[2 * e + 1]);
      lo += 0x7fffu + ((lo >> 16) & 1u);
      hi += 0x7fffu + ((hi >> 16) & 1u);
      o[e] = (lo >> 16) | (hi & 0xffff0000u);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(acc[e]);
  }
  out[(b * S + s) * vec_per_row + v] = o;
}

// Column sums of a SHORT matrix (hundreds to a few thousand rows: the bias gradients of the Focused Decoder's
// 1080-row linears, the per-wave partials of the token kernels) in ONE launch: a workgroup owns one 16-byte column
// group (8 bf16 / 4 fp32 columns) and walks all rows, 256 at a time, four loads in flight per thread; the 256 row lanes
// meet in LDS.  The 16-byte pieces of a wave are a row pitch apart, but the neighbouring column groups' workgroups
// read the rest of the same lines at the same time: the matrix (<= a few MB) is served by L2.  torch's reduce kernel
// needs 13-25 us for these shapes (two passes, 64-byte accesses); this one is latency of ~(rows / 1024) load rounds.
template <bool BF16>
__global__ __launch_bounds__(256) void colsum_small(const u32x4r* __restrict__ x, float* __restrict__ out, long rows,
                                                    int vec_per_row) {
  constexpr int NE = BF16 ? 8 : 4;
  __shared__ float red[256][NE + 1];
  const int v = blockIdx.x;
  float acc[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) acc[e] = 0.f;
  auto add = [&](const u32x4r& q) {
    if (BF16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[2 * e] += __uint_as_float(q[e] << 16);
        acc[2 * e + 1] += __uint_as_float(q[e] & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += __uint_as_float(q[e]);
    }
  };
  long r = threadIdx.x;
  for (; r + 768 < rows; r += 1024) {
    const u32x4r q0 = x[r * vec_per_row + v], q1 = x[(r + 256) * vec_per_row + v];
    const u32x4r q2 = x[(r + 512) * vec_per_row + v], q3 = x[(r + 768) * vec_per_row + v];
    add(q0); add(q1); add(q2); add(q3);
  }
  for (; r < rows; r += 256) add(x[r * vec_per_row + v]);
#pragma unroll
  for (int e = 0; e < NE; ++e) red[threadIdx.x][e] = acc[e];
  __syncthreads();
  for (int half = 128; half >= NE; half >>= 1) {          // tree over the row lanes, down to NE rows
    if (static_cast<int>(threadIdx.x) < half) {
#pragma unroll
      for (int e = 0; e < NE; ++e) red[threadIdx.x][e] += red[threadIdx.x + half][e];
    }
    __syncthreads();
  }
  if (static_cast<int>(threadIdx.x) < NE) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < NE; ++k) t += red[k][threadIdx.x];
    out[static_cast<long>(v) * NE + threadIdx.x] = t;
  }
}

// Column sums of a bf16 (rows, cols) matrix in fp32: the bias gradients of the token projections and of the FPN
// output convolutions (channels-last: rows = voxels).  Pass 1: kColsumBlocks workgroups, a thread owns one 16-byte
// column group and every (blocks * R)-th row, the R row lanes of a workgroup meet in LDS; pass 2 adds the
// per-workgroup partials.  torch's sum(0) runs at 2 TB/s on the 234 000 x 384 matrix.
constexpr int kColsumBlocks = 1024;

__global__ __launch_bounds__(256) void colsum_partial(const u32x4r* __restrict__ x, float* __restrict__ partials,
                                                      long rows, int vec_per_row, int rows_per_block) {
  __shared__ float sh[256 * 8];
  const int tid = threadIdx.x;
  const int rr = tid / vec_per_row, v = tid - rr * vec_per_row;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (rr < rows_per_block) {
    for (long r = static_cast<long>(blockIdx.x) * rows_per_block + rr; r < rows;
         r += static_cast<long>(kColsumBlocks) * rows_per_block) {
      const u32x4r q = x[r * vec_per_row + v];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[2 * e] += __uint_as_float(q[e] << 16);
        acc[2 * e + 1] += __uint_as_float(q[e] & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) sh[tid * 8 + e] = acc[e];
  __syncthreads();
  if (tid < vec_per_row) {
    for (int o = 1; o < rows_per_block; ++o)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += sh[(o * vec_per_row + tid) * 8 + e];
    float* dst = partials + (static_cast<long>(blockIdx.x) * vec_per_row + tid) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = acc[e];
  }
}

// 32 columns per workgroup, 32 row lanes of 32 partials each, the 32 sums meet in LDS
__global__ __launch_bounds__(1024) void colsum_final(const float* __restrict__ partials, float* __restrict__ out, int cols) {
  // 32 groups of 32 columns: a thread adds kColsumBlocks / 32 partial rows, eight loads in flight at a time (with 8
  // groups and a serial chain of 128 dependent L2 round trips this pass took 35 us -- as long as the reduction itself)
  __shared__ float sh[32][32];
  const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < cols) {
    const float* p = partials + static_cast<long>(grp) * cols + c;
    const long step = 32L * cols;
    int b = grp;
    for (; b + 96 < kColsumBlocks; b += 128) {
      a0 += p[0]; a1 += p[step]; a2 += p[2 * step]; a3 += p[3 * step];
      p += 4 * step;
    }
    for (; b < kColsumBlocks; b += 32) {
      a0 += *p;
      p += step;
    }
  }
  sh[grp][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && c < cols) {
    float a = 0.f;
#pragma unroll
    for (int g = 0; g < 32; ++g) a += sh[g][cl];
    out[c] = a;
  }
}

}  // namespace transoar

using namespace transoar;

extern "C" int transoar_rows_gather(const void* src, const int* index, void* out, int B, long S, long K,
                                    int row_bytes, void* hip_stream) {
  if (!src || !index || !out) return -1;
  if (B <= 0 || S <= 0 || K <= 0 || row_bytes <= 0 || (row_bytes & 15)) return -2;
  const int vpr = row_bytes / 16;
  const dim3 grid(static_cast<unsigned>((K * vpr + 255) / 256), static_cast<unsigned>(B));
  hipLaunchKernelGGL(rows_gather, grid, dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                     static_cast<const u32x4r*>(src), index, static_cast<u32x4r*>(out), S, K, vpr);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_rows_gather_axpy(const void* src, const int* index, const float* scale, const void* resid, void* out,
                                         int B, long S, long K, int row_bytes, void* hip_stream) {
  if (!src || !index || !out) return -1;
  if (B <= 0 || S <= 0 || K <= 0 || row_bytes <= 0 || (row_bytes & 15)) return -2;
  const int vpr = row_bytes / 16;
  const dim3 grid(static_cast<unsigned>((K * vpr + 255) / 256), static_cast<unsigned>(B));
  hipLaunchKernelGGL(rows_gather_axpy, grid, dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                     static_cast<const u32x4r*>(src), index, scale, static_cast<const u32x4r*>(resid),
                     static_cast<u32x4r*>(out), S, K, vpr);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_rows_pull_sum(const void* g, const int* inv_ptr, const int* inv_idx, void* out, int B,
                                      long S, long K, int row_bytes, int is_bf16, void* hip_stream) {
  if (!g || !inv_ptr || !inv_idx || !out) return -1;
  if (B <= 0 || S <= 0 || K <= 0 || row_bytes <= 0 || (row_bytes & 15)) return -2;
  const int vpr = row_bytes / 16;
  const dim3 grid(static_cast<unsigned>((S * vpr + 255) / 256), static_cast<unsigned>(B));
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  if (is_bf16)
    hipLaunchKernelGGL(rows_pull_sum<true>, grid, dim3(256), 0, st, static_cast<const u32x4r*>(g), inv_ptr, inv_idx,
                       static_cast<u32x4r*>(out), S, K, vpr);
  else
    hipLaunchKernelGGL(rows_pull_sum<false>, grid, dim3(256), 0, st, static_cast<const u32x4r*>(g), inv_ptr, inv_idx,
                       static_cast<u32x4r*>(out), S, K, vpr);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_rows_colsum_small(const void* x, float* out, long rows, int cols, int is_bf16, void* hip_stream) {
  using namespace transoar;
  if (!x || !out) return -1;
  const int per = is_bf16 ? 8 : 4;
  if (rows <= 0 || cols <= 0 || (cols % per) || rows > (1L << 20)) return -2;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const int vpr = cols / per;
  if (is_bf16)
    hipLaunchKernelGGL(colsum_small<true>, dim3(vpr), dim3(256), 0, st, static_cast<const u32x4r*>(x), out, rows, vpr);
  else
    hipLaunchKernelGGL(colsum_small<false>, dim3(vpr), dim3(256), 0, st, static_cast<const u32x4r*>(x), out, rows, vpr);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_rows_colsum_workspace_floats(int cols) { return transoar::kColsumBlocks * cols; }

extern "C" int transoar_rows_colsum(const void* x, float* out, float* workspace, long rows, int cols, void* hip_stream) {
  using namespace transoar;
  if (!x || !out || !workspace) return -1;
  if (rows <= 0 || cols <= 0 || (cols & 7) || cols / 8 > 256) return -2;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const int vpr = cols / 8, rpb = 256 / vpr;
  hipLaunchKernelGGL(colsum_partial, dim3(kColsumBlocks), dim3(256), 0, st, static_cast<const u32x4r*>(x), workspace, rows,
                     vpr, rpb);
  hipLaunchKernelGGL(colsum_final, dim3((cols + 31) / 32), dim3(1024), 0, st, workspace, out, cols);
  return static_cast<int>(hipGetLastError());
}
