// InstanceNorm3d(affine) + ReLU on channels-last bf16 activations for gfx950.
//
// Replaces the cuDNN/MIOpen batch-norm kernels PyTorch runs for the
// nn.InstanceNorm3d + nn.ReLU pairs of the reference's encoder blocks
// (transoar/models/backbones/encoder_blocks.py:34-36, 44-46).  HBM-bound:
// forward = one read for the statistics + one read/one write to apply;
// backward = one read of (x, dy) for the two reductions + one read/one write
// for dx.  The ReLU is folded in (its mask is recomputed from x, nothing extra
// is stored).  Statistics are fp32 per thread, fp64 across threads.
//
// Layout: x (N, V, C) bf16 with V = D*H*W voxels, C % 8 == 0.  A thread moves
// 16 bytes (8 channels of one voxel); the block size (192) is a multiple of
// C/8 for every width of the backbone (24..768 = 3*2^k*8), so a thread keeps
// the same 8 channels for its whole strided walk over the voxels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/transoar_instnorm.h"

namespace transoar {

using u32x4n = __attribute__((ext_vector_type(4))) unsigned int;
constexpr int kINThreads = 192;
constexpr int kINDepth = 4;      // 16-byte loads in flight per thread
constexpr int kINBlocksPerCU = 4;  // 8 measured no faster
constexpr int kINDepthDx = 2;    // the dx pass also stores: two x / dy pairs in flight

__device__ __forceinline__ void unpack8(const u32x4n& r, float (&o)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = __uint_as_float(r[i] << 16);
    o[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ unsigned short f2bf_n(float f) {
  return __builtin_bit_cast(unsigned short, static_cast<__bf16>(f));      // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
}
__device__ __forceinline__ unsigned pack2_bf16_n(float a, float b) {        // one v_cvt_pk_bf16_f32 (round to nearest even)
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
__device__ __forceinline__ u32x4n pack8(const float (&o)[8]) {
  u32x4n r;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r[i] = pack2_bf16_n(o[2 * i], o[2 * i + 1]);
  return r;
}

// Sum the per-thread partials a[0..K) of all threads that share this thread's
// channel chunk and add them (fp64 atomics) to dst[chunk*8 + e][k].
template <int K>
__device__ __forceinline__ void block_reduce_to_global(const float (&a)[K][8], int chunks, double* dst,
                                                       int dst_stride) {
  __shared__ float sm[kINThreads][K * 8 + 1];
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) sm[tid][k * 8 + e] = a[k][e];
  __syncthreads();
  // thread t < chunks*K*8 finishes one (chunk, k, e) column
  const int cols = chunks * K * 8;
  for (int c = tid; c < cols; c += kINThreads) {
    const int chunk = c / (K * 8), ke = c - chunk * (K * 8);
    double s = 0.0;
    for (int t = chunk; t < kINThreads; t += chunks) s += static_cast<double>(sm[t][ke]);
    const int k = ke >> 3, e = ke & 7;
    __hip_atomic_fetch_add(dst + static_cast<long>(chunk * 8 + e) * dst_stride + k, s, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  }
}

// stats[n][c][0] += sum x ; stats[n][c][1] += sum x^2        (fp64, pre-zeroed)
__global__ __launch_bounds__(kINThreads) void instnorm_stats(const unsigned short* __restrict__ x,
                                                             double* __restrict__ stats, long V, int C,
                                                             int blocks_per_sample) {
  const int chunks = C >> 3;
  const int n = blockIdx.x / blocks_per_sample, bi = blockIdx.x % blocks_per_sample;
  const int chunk = threadIdx.x % chunks;
  const long vstep = kINThreads / chunks;
  const long v_per_block = (V + blocks_per_sample - 1) / blocks_per_sample;
  const long v0 = bi * v_per_block, v1 = min(V, v0 + v_per_block);
  const unsigned short* xs = x + static_cast<long>(n) * V * C;
  float acc[2][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[0][e] = acc[1][e] = 0.f;
  auto take = [&](const u32x4n& p) {
    float f[8];
    unpack8(p, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      acc[0][e] += f[e];
      acc[1][e] += f[e] * f[e];
    }
  };
  // kINDepth loads in flight per thread (one at a time left these kernels at 40-60 % of the HBM rate); same order of sums
  long v = v0 + threadIdx.x / chunks;
  for (; v + (kINDepth - 1) * vstep < v1; v += kINDepth * vstep) {
    u32x4n p[kINDepth];
#pragma unroll
    for (int i = 0; i < kINDepth; ++i) p[i] = *reinterpret_cast<const u32x4n*>(xs + (v + i * vstep) * C + chunk * 8);
#pragma unroll
    for (int i = 0; i < kINDepth; ++i) take(p[i]);
  }
  for (; v < v1; v += vstep) take(*reinterpret_cast<const u32x4n*>(xs + v * C + chunk * 8));
  block_reduce_to_global<2>(acc, chunks, stats + static_cast<long>(n) * C * 2, 2);
}

// stats[n][c][which] += sum over the rows of part[n * rows_per_sample + r][which][32 channels]   (fp64, pre-zeroed):
// the per-workgroup partial sums conv3d_k3_lds<.., STATS> leaves behind (csrc/conv3d.hip).  grid (N, kPartBlocks).
constexpr int kPartBlocks = 8;
__global__ __launch_bounds__(256) void instnorm_stats_from_parts(const float* __restrict__ part, int rows_per_sample, int C,
                                                                 double* __restrict__ stats) {
  __shared__ double sh[4][64];
  const int n = blockIdx.x, e = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const float* p = part + static_cast<long>(n) * rows_per_sample * 64 + e;
  double acc = 0.0;
  for (int r = blockIdx.y * 4 + grp; r < rows_per_sample; r += 4 * kPartBlocks) acc += static_cast<double>(p[static_cast<long>(r) * 64]);
  sh[grp][e] = acc;
  __syncthreads();
  if (grp == 0) {
    const int which = e >> 5, c = e & 31;
    if (c < C) atomicAdd(stats + (static_cast<long>(n) * C + c) * 2 + which, sh[0][e] + sh[1][e] + sh[2][e] + sh[3][e]);
  }
}

// y = relu((x - mean) * rstd * gamma + beta)
__global__ __launch_bounds__(kINThreads) void instnorm_apply_relu(
    const unsigned short* __restrict__ x, const double* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta, unsigned short* __restrict__ y, float* __restrict__ mean_rstd, long V,
    int C, float eps, int blocks_per_sample, int relu) {
  const int chunks = C >> 3;
  const int n = blockIdx.x / blocks_per_sample, bi = blockIdx.x % blocks_per_sample;
  const int chunk = threadIdx.x % chunks;
  const long vstep = kINThreads / chunks;
  const long v_per_block = (V + blocks_per_sample - 1) / blocks_per_sample;
  const long v0 = bi * v_per_block, v1 = min(V, v0 + v_per_block);
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = chunk * 8 + e;
    const double s = stats[(static_cast<long>(n) * C + c) * 2], ss = stats[(static_cast<long>(n) * C + c) * 2 + 1];
    const double m = s / static_cast<double>(V);
    double var = ss / static_cast<double>(V) - m * m;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
    sc[e] = rstd * gamma[c];
    sh[e] = beta[c] - static_cast<float>(m) * sc[e];
    if (bi == 0 && threadIdx.x < chunks) {
      mean_rstd[(static_cast<long>(n) * C + c) * 2] = static_cast<float>(m);
      mean_rstd[(static_cast<long>(n) * C + c) * 2 + 1] = rstd;
    }
  }
  const long base = static_cast<long>(n) * V * C;
  auto apply = [&](const u32x4n& p, long vv) {
    float f[8];
    unpack8(p, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      f[e] = f[e] * sc[e] + sh[e];
      if (relu) f[e] = f[e] > 0.f ? f[e] : 0.f;
    }
    *reinterpret_cast<u32x4n*>(y + base + vv * C + chunk * 8) = pack8(f);
  };
  long v = v0 + threadIdx.x / chunks;
  for (; v + (kINDepth - 1) * vstep < v1; v += kINDepth * vstep) {
    u32x4n p[kINDepth];
#pragma unroll
    for (int i = 0; i < kINDepth; ++i) p[i] = *reinterpret_cast<const u32x4n*>(x + base + (v + i * vstep) * C + chunk * 8);
#pragma unroll
    for (int i = 0; i < kINDepth; ++i) apply(p[i], v + i * vstep);
  }
  for (; v < v1; v += vstep) apply(*reinterpret_cast<const u32x4n*>(x + base + v * C + chunk * 8), v);
}

// red[n][c][0] += sum g ; red[n][c][1] += sum g*xhat,  g = dy * [relu active]
__global__ __launch_bounds__(kINThreads) void instnorm_bwd_reduce(
    const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy,
    const float* __restrict__ mean_rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    double* __restrict__ red, long V, int C, int blocks_per_sample, int relu) {
  const int chunks = C >> 3;
  const int n = blockIdx.x / blocks_per_sample, bi = blockIdx.x % blocks_per_sample;
  const int chunk = threadIdx.x % chunks;
  const long vstep = kINThreads / chunks;
  const long v_per_block = (V + blocks_per_sample - 1) / blocks_per_sample;
  const long v0 = bi * v_per_block, v1 = min(V, v0 + v_per_block);
  float m[8], rs[8], ga[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = chunk * 8 + e;
    m[e] = mean_rstd[(static_cast<long>(n) * C + c) * 2];
    rs[e] = mean_rstd[(static_cast<long>(n) * C + c) * 2 + 1];
    ga[e] = gamma[c];
    be[e] = beta[c];
  }
  const long base = static_cast<long>(n) * V * C;
  float acc[2][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[0][e] = acc[1][e] = 0.f;
  auto take = [&](const u32x4n& px, const u32x4n& pg) {
    float f[8], g[8];
    unpack8(px, f);
    unpack8(pg, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (f[e] - m[e]) * rs[e];
      const float gg = (!relu || xh * ga[e] + be[e] > 0.f) ? g[e] : 0.f;
      acc[0][e] += gg;
      acc[1][e] += gg * xh;
    }
  };
  long v = v0 + threadIdx.x / chunks;
  for (; v + (kINDepth - 1) * vstep < v1; v += kINDepth * vstep) {
    u32x4n px[kINDepth], pg[kINDepth];
#pragma unroll
    for (int i = 0; i < kINDepth; ++i) {
      px[i] = *reinterpret_cast<const u32x4n*>(x + base + (v + i * vstep) * C + chunk * 8);
      pg[i] = *reinterpret_cast<const u32x4n*>(dy + base + (v + i * vstep) * C + chunk * 8);
    }
#pragma unroll
    for (int i = 0; i < kINDepth; ++i) take(px[i], pg[i]);
  }
  for (; v < v1; v += vstep)
    take(*reinterpret_cast<const u32x4n*>(x + base + v * C + chunk * 8), *reinterpret_cast<const u32x4n*>(dy + base + v * C + chunk * 8));
  block_reduce_to_global<2>(acc, chunks, red + static_cast<long>(n) * C * 2, 2);
}

// dx = rstd*gamma*(g - mean(g) - xhat*mean(g*xhat))
__global__ __launch_bounds__(kINThreads) void instnorm_bwd_dx(
    const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy,
    const float* __restrict__ mean_rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const double* __restrict__ red, unsigned short* __restrict__ dx, long V, int C, int blocks_per_sample,
    int relu, float* __restrict__ dparams, int N) {
  const int chunks = C >> 3;
  const int n = blockIdx.x / blocks_per_sample, bi = blockIdx.x % blocks_per_sample;
  if (dparams && blockIdx.x == 0) {
    // the parameter gradients on the side: dbeta_c = sum_n red[n][c][0], dgamma_c = sum_n red[n][c][1] (a torch sum and two
    // fp64 -> fp32 casts per layer otherwise: 36 launches of the step)
    for (int i = threadIdx.x; i < 2 * C; i += kINThreads) {
      const int k = i / C, c = i - k * C;
      double s = 0.0;
      for (int nn = 0; nn < N; ++nn) s += red[(static_cast<long>(nn) * C + c) * 2 + k];
      dparams[i] = static_cast<float>(s);
    }
  }
  const int chunk = threadIdx.x % chunks;
  const long vstep = kINThreads / chunks;
  const long v_per_block = (V + blocks_per_sample - 1) / blocks_per_sample;
  const long v0 = bi * v_per_block, v1 = min(V, v0 + v_per_block);
  float m[8], rs[8], ga[8], be[8], mg[8], mgx[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = chunk * 8 + e;
    const long k = static_cast<long>(n) * C + c;
    m[e] = mean_rstd[k * 2];
    rs[e] = mean_rstd[k * 2 + 1];
    ga[e] = gamma[c];
    be[e] = beta[c];
    mg[e] = static_cast<float>(red[k * 2] / static_cast<double>(V));
    mgx[e] = static_cast<float>(red[k * 2 + 1] / static_cast<double>(V));
  }
  const long base = static_cast<long>(n) * V * C;
  auto put = [&](const u32x4n& px, const u32x4n& pg, long vv) {
    float f[8], g[8];
    unpack8(px, f);
    unpack8(pg, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (f[e] - m[e]) * rs[e];
      const float gg = (!relu || xh * ga[e] + be[e] > 0.f) ? g[e] : 0.f;
      f[e] = rs[e] * ga[e] * (gg - mg[e] - xh * mgx[e]);
    }
    *reinterpret_cast<u32x4n*>(dx + base + vv * C + chunk * 8) = pack8(f);
  };
  long v = v0 + threadIdx.x / chunks;
  for (; v + (kINDepthDx - 1) * vstep < v1; v += kINDepthDx * vstep) {
    u32x4n px[kINDepthDx], pg[kINDepthDx];
#pragma unroll
    for (int i = 0; i < kINDepthDx; ++i) {
      px[i] = *reinterpret_cast<const u32x4n*>(x + base + (v + i * vstep) * C + chunk * 8);
      pg[i] = *reinterpret_cast<const u32x4n*>(dy + base + (v + i * vstep) * C + chunk * 8);
    }
#pragma unroll
    for (int i = 0; i < kINDepthDx; ++i) put(px[i], pg[i], v + i * vstep);
  }
  for (; v < v1; v += vstep)
    put(*reinterpret_cast<const u32x4n*>(x + base + v * C + chunk * 8), *reinterpret_cast<const u32x4n*>(dy + base + v * C + chunk * 8), v);
}

static int pick_blocks(long V, int N, int C) {
  // ~4 blocks per CU overall; a thread should still have >= 8 of the 16-byte pieces (V * C/8 per sample)
  // to walk -- the deep encoder stages have a few hundred voxels per sample, and one block per sample
  // (the old "at least 4096 voxels per block") left them to 2 workgroups: 0.1-0.3 ms per call
  long per = (kINBlocksPerCU * 256L + N - 1) / N;
  const long cap = (V * (C >> 3) + 8L * kINThreads - 1) / (8L * kINThreads);
  if (per > cap) per = cap;
  return static_cast<int>(per < 1 ? 1 : per);
}

static int check(const void* a, const void* b, int N, long V, int C) {
  if (!a || !b) return TRANSOAR_IN_ERR_NULL;
  if (N <= 0 || V <= 0 || C <= 0) return TRANSOAR_IN_ERR_DIM;
  if ((C & 7) || (kINThreads % (C >> 3))) return TRANSOAR_IN_ERR_CHANNELS;
  return 0;
}

}  // namespace transoar

using namespace transoar;

extern "C" int transoar_instnorm_relu_forward(const void* x, const float* gamma, const float* beta, void* y,
                                              double* stats_ws, float* mean_rstd, int N, long V, int C, float eps,
                                              int relu, void* hip_stream) {
  const int rc = check(x, y, N, V, C);
  if (rc) return rc;
  if (!gamma || !beta || !stats_ws || !mean_rstd) return TRANSOAR_IN_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const int bps = pick_blocks(V, N, C);
  hipError_t e = hipMemsetAsync(stats_ws, 0, sizeof(double) * 2 * N * C, st);
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL(instnorm_stats, dim3(N * bps), dim3(kINThreads), 0, st, static_cast<const unsigned short*>(x),
                     stats_ws, V, C, bps);
  hipLaunchKernelGGL(instnorm_apply_relu, dim3(N * bps), dim3(kINThreads), 0, st,
                     static_cast<const unsigned short*>(x), stats_ws, gamma, beta, static_cast<unsigned short*>(y),
                     mean_rstd, V, C, eps, bps, relu);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_instnorm_relu_forward_parts(const void* x, const float* gamma, const float* beta, void* y,
                                                    const float* stat_part, int rows_per_sample, double* stats_ws,
                                                    float* mean_rstd, int N, long V, int C, float eps, int relu,
                                                    void* hip_stream) {
  const int rc = check(x, y, N, V, C);
  if (rc) return rc;
  if (!gamma || !beta || !stats_ws || !mean_rstd || !stat_part) return TRANSOAR_IN_ERR_NULL;
  if (C > 32 || rows_per_sample <= 0) return TRANSOAR_IN_ERR_DIM;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const int bps = pick_blocks(V, N, C);
  hipError_t e = hipMemsetAsync(stats_ws, 0, sizeof(double) * 2 * N * C, st);
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL(instnorm_stats_from_parts, dim3(N, kPartBlocks), dim3(256), 0, st, stat_part, rows_per_sample, C, stats_ws);
  hipLaunchKernelGGL(instnorm_apply_relu, dim3(N * bps), dim3(kINThreads), 0, st,
                     static_cast<const unsigned short*>(x), stats_ws, gamma, beta, static_cast<unsigned short*>(y),
                     mean_rstd, V, C, eps, bps, relu);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_instnorm_relu_backward(const void* x, const void* dy, const float* gamma, const float* beta,
                                               const float* mean_rstd, void* dx, double* red_ws, float* dparams, int N, long V,
                                               int C, int relu, void* hip_stream) {
  const int rc = check(x, dy, N, V, C);
  if (rc) return rc;
  if (!gamma || !beta || !mean_rstd || !dx || !red_ws) return TRANSOAR_IN_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const int bps = pick_blocks(V, N, C);
  hipError_t e = hipMemsetAsync(red_ws, 0, sizeof(double) * 2 * N * C, st);
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL(instnorm_bwd_reduce, dim3(N * bps), dim3(kINThreads), 0, st,
                     static_cast<const unsigned short*>(x), static_cast<const unsigned short*>(dy), mean_rstd, gamma,
                     beta, red_ws, V, C, bps, relu);
  hipLaunchKernelGGL(instnorm_bwd_dx, dim3(N * bps), dim3(kINThreads), 0, st, static_cast<const unsigned short*>(x),
                     static_cast<const unsigned short*>(dy), mean_rstd, gamma, beta, red_ws,
                     static_cast<unsigned short*>(dx), V, C, bps, relu, dparams, N);
  return static_cast<int>(hipGetLastError());
}

extern "C" int transoar_instnorm_abi_version(void) { return 3; }
