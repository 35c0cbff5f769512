// Stand-alone timing harness for conv3d_c1_fwd variants (dev tool; hipcc, no torch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/c1h tools/harness/c1_harness.hip && /tmp/c1h
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define main_conv3d_skip 1
#include "../../transoar_amd/csrc/conv3d.hip"

using namespace transoar;

template <int MODE>
__global__ __launch_bounds__(256) void c1_variant(const unsigned short* __restrict__ x, const float* __restrict__ w,
                                                  unsigned short* __restrict__ y, int N, int D, int H, int W, int Cout,
                                                  long n_vox) {
  __shared__ u32x4c stage[256 * 24 / 8];
  const long v0 = static_cast<long>(blockIdx.x) * 256;
  const long v = v0 + threadIdx.x;
  const long vc = v < n_vox ? v : n_vox - 1;
  const int ow = static_cast<int>(vc % W);
  const long r1 = vc / W;
  const int oh = static_cast<int>(r1 % H);
  const long r2 = r1 / H;
  const int od = static_cast<int>(r2 % D);
  const long nbase = (r2 / D) * D;
  float xv[27];
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int id = od + kd - 1, ih = oh + kh - 1, iw = ow + kw - 1;
        const bool ok = (unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        const int cd = min(max(id, 0), D - 1), chh = min(max(ih, 0), H - 1), cw = min(max(iw, 0), W - 1);
        float val = MODE == 1 ? 1.0f + kd : bf2f(x[((nbase + cd) * H + chh) * W + cw]);
        xv[kd * 9 + kh * 3 + kw] = ok ? val : 0.f;
      }
  const int cpv = Cout >> 3;
  for (int c0 = 0; c0 < Cout; c0 += 8) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int t = 0; t < (MODE == 2 ? 1 : 27); ++t) {
      const float* wt = w + t * Cout + c0;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += xv[t] * wt[e];
    }
    u32x4c pk;
#pragma unroll
    for (int e = 0; e < 4; ++e) pk[e] = (unsigned)f2bf(acc[2 * e]) | ((unsigned)f2bf(acc[2 * e + 1]) << 16);
    stage[threadIdx.x * cpv + (c0 >> 3)] = pk;
  }
  __syncthreads();
  if (MODE == 3) { if (stage[threadIdx.x][0] == 0x12345678u) y[v0] = 1; return; }
  const long chunks = min(256L, n_vox - v0) * cpv;
  u32x4c* dst = reinterpret_cast<u32x4c*>(y + v0 * Cout);
  for (int i = threadIdx.x; i < chunks; i += 256) dst[i] = stage[i];
}


// 4 consecutive voxels along W per thread: every scalar-loaded weight group feeds 4x the FMAs
template <int VPT, bool STAGE>
__global__ __launch_bounds__(256) void c1_multi(const unsigned short* __restrict__ x, const float* __restrict__ w,
                                                unsigned short* __restrict__ y, int N, int D, int H, int W, int Cout,
                                                long n_groups) {
  const long gidx = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;     // group of VPT voxels
  if (gidx >= n_groups) return;
  const int WG = W / VPT;
  const int gw = static_cast<int>(gidx % WG);
  const long r1 = gidx / WG;
  const int oh = static_cast<int>(r1 % H);
  const long r2 = r1 / H;
  const int od = static_cast<int>(r2 % D);
  const long nbase = (r2 / D) * D;
  const int ow0 = gw * VPT;
  float xv[9][VPT + 2];
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int id = od + kd - 1, ih = oh + kh - 1;
      const bool okr = (unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H;
      const int cd = min(max(id, 0), D - 1), chh = min(max(ih, 0), H - 1);
      const unsigned short* row = x + ((nbase + cd) * H + chh) * W;
#pragma unroll
      for (int j = 0; j < VPT + 2; ++j) {
        const int iw = ow0 + j - 1;
        const bool ok = okr && (unsigned)iw < (unsigned)W;
        const float val = bf2f(row[min(max(iw, 0), W - 1)]);
        xv[kd * 3 + kh][j] = ok ? val : 0.f;
      }
    }
  const long v0 = ((nbase + od) * H + oh) * W + ow0;
  for (int c0 = 0; c0 < Cout; c0 += 8) {
    float acc[VPT][8];
#pragma unroll
    for (int p = 0; p < VPT; ++p)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[p][e] = 0.f;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const float* wt = w + t * Cout + c0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float we = wt[e];
#pragma unroll
        for (int p = 0; p < VPT; ++p) acc[p][e] += xv[t / 3][p + t % 3] * we;
      }
    }
#pragma unroll
    for (int p = 0; p < VPT; ++p) {
      u32x4c pk;
#pragma unroll
      for (int e = 0; e < 4; ++e) pk[e] = (unsigned)f2bf(acc[p][2 * e]) | ((unsigned)f2bf(acc[p][2 * e + 1]) << 16);
      *reinterpret_cast<u32x4c*>(y + (v0 + p) * Cout + c0) = pk;
    }
  }
}


// tap-outer: all Cout accumulators live, one weight row (Cout floats) per tap; MODE 0: scalar loads +
// sched barrier per tap, MODE 1: weights in LDS (broadcast reads)
template <int COUT, int MODE>
__global__ __launch_bounds__(256) void c1_tapouter(const unsigned short* __restrict__ x, const float* __restrict__ w,
                                                   unsigned short* __restrict__ y, int N, int D, int H, int W, int Cout,
                                                   long n_vox) {
  __shared__ float wsh[27 * COUT];
  if (MODE == 1) {
    for (int i = threadIdx.x; i < 27 * COUT; i += 256) wsh[i] = w[i];
    __syncthreads();
  }
  const long v = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (v >= n_vox) return;
  const int ow = static_cast<int>(v % W);
  const long r1 = v / W;
  const int oh = static_cast<int>(r1 % H);
  const long r2 = r1 / H;
  const int od = static_cast<int>(r2 % D);
  const long nbase = (r2 / D) * D;
  float xv[27];
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int id = od + kd - 1, ih = oh + kh - 1, iw = ow + kw - 1;
        const bool ok = (unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        const int cd = min(max(id, 0), D - 1), chh = min(max(ih, 0), H - 1), cw = min(max(iw, 0), W - 1);
        const float val = bf2f(x[((nbase + cd) * H + chh) * W + cw]);
        xv[kd * 9 + kh * 3 + kw] = ok ? val : 0.f;
      }
  float acc[COUT];
#pragma unroll
  for (int e = 0; e < COUT; ++e) acc[e] = 0.f;
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    if (MODE == 1) {
#pragma unroll
      for (int e4 = 0; e4 < COUT; e4 += 4) {
        const float4 wv = *reinterpret_cast<const float4*>(&wsh[t * COUT + e4]);
        acc[e4] += xv[t] * wv.x; acc[e4 + 1] += xv[t] * wv.y; acc[e4 + 2] += xv[t] * wv.z; acc[e4 + 3] += xv[t] * wv.w;
      }
    } else {
      const float* wt = w + t * COUT;
#pragma unroll
      for (int e = 0; e < COUT; ++e) acc[e] += xv[t] * wt[e];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int c0 = 0; c0 < COUT; c0 += 8) {
    u32x4c pk;
#pragma unroll
    for (int e = 0; e < 4; ++e) pk[e] = (unsigned)f2bf(acc[c0 + 2 * e]) | ((unsigned)f2bf(acc[c0 + 2 * e + 1]) << 16);
    *reinterpret_cast<u32x4c*>(y + v * COUT + c0) = pk;
  }
}

int main() {
  const int N = 2, D = 160, H = 160, W = 256, Cout = 24;
  const long n_vox = (long)N * D * H * W;
  unsigned short *x, *y; float* w;
  hipMalloc(&x, n_vox * 2); hipMalloc(&y, n_vox * Cout * 2); hipMalloc(&w, 27 * Cout * 4);
  hipMemset(x, 0x3f, n_vox * 2); hipMemset(w, 0, 27 * Cout * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](const char* name, auto kern) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3((n_vox + 255) / 256), dim3(256), 0, 0, x, w, y, N, D, H, W, Cout, n_vox);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3((n_vox + 255) / 256), dim3(256), 0, 0, x, w, y, N, D, H, W, Cout, n_vox);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s %.3f ms  (%s)\n", name, ms / 5, hipGetErrorString(hipGetLastError()));
  };
  run("shipped conv3d_c1_fwd", conv3d_c1_fwd);
  run("variant same", c1_variant<0>);
  run("no x loads", c1_variant<1>);
  run("1 tap of 27", c1_variant<2>);
  run("no global stores", c1_variant<3>);
  auto run2 = [&](const char* name, auto kern, int vpt) {
    const long ng = n_vox / vpt;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3((ng + 255) / 256), dim3(256), 0, 0, x, w, y, N, D, H, W, Cout, ng);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3((ng + 255) / 256), dim3(256), 0, 0, x, w, y, N, D, H, W, Cout, ng);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s %.3f ms  (%s)\n", name, ms / 5, hipGetErrorString(hipGetLastError()));
  };
  run("tap-outer, s_load + sched barrier", c1_tapouter<24, 0>);
  run("tap-outer, LDS weights", c1_tapouter<24, 1>);
  run2("2 voxels/thread", c1_multi<2, false>, 2);
  run2("4 voxels/thread", c1_multi<4, false>, 4);
  run2("8 voxels/thread", c1_multi<8, false>, 8);
  return 0;
}
