// Stand-alone check + timing of conv3d_k3_wgrad_lds (dev tool; hipcc, no torch).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "../../transoar_amd/csrc/conv3d.hip"
using namespace transoar;

static unsigned short f2b(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float b2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
  {  // correctness on a small volume against a scalar reference
    const int N = 2, D = 3, H = 4, W = 64, Cin = 24, Cout = 24, n_wg = 5;
    const long V = (long)N * D * H * W;
    std::vector<unsigned short> hx(V * Cin), hdy(V * Cout);
    srand(1);
    for (auto& v : hx) v = f2b((rand() % 2001 - 1000) / 1000.0f);
    for (auto& v : hdy) v = f2b((rand() % 2001 - 1000) / 1000.0f);
    std::vector<double> ref(27 * Cout * Cin, 0.0);
    for (int n = 0; n < N; ++n) for (int d = 0; d < D; ++d) for (int h = 0; h < H; ++h) for (int w = 0; w < W; ++w) {
      const long v = ((long)(n * D + d) * H + h) * W + w;
      for (int kd = 0; kd < 3; ++kd) for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw) {
        const int id = d + kd - 1, ih = h + kh - 1, iw = w + kw - 1;
        if (id < 0 || id >= D || ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
        const long vi = ((long)(n * D + id) * H + ih) * W + iw;
        const int tap = (kd * 3 + kh) * 3 + kw;
        for (int co = 0; co < Cout; ++co) for (int ci = 0; ci < Cin; ++ci)
          ref[(tap * Cout + co) * Cin + ci] += (double)b2f(hdy[v * Cout + co]) * b2f(hx[vi * Cin + ci]);
      }
    }
    unsigned short *x, *dy; float* part;
    hipMalloc(&x, hx.size() * 2); hipMalloc(&dy, hdy.size() * 2); hipMalloc(&part, (size_t)n_wg * 27 * 1024 * 4);
    hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dy, hdy.data(), hdy.size() * 2, hipMemcpyHostToDevice);
    int rc = transoar_conv3d_k3_wgrad_lds(x, dy, part, n_wg, N, D, H, W, Cin, Cout, 0, Cin, 0, Cout, nullptr);
    hipDeviceSynchronize();
    std::vector<float> hp((size_t)n_wg * 27 * 1024);
    hipMemcpy(hp.data(), part, hp.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int tap = 0; tap < 27; ++tap) for (int co = 0; co < 32; ++co) for (int ci = 0; ci < 32; ++ci) {
      double got = 0; for (int g = 0; g < n_wg; ++g) got += hp[((size_t)(g * 27 + tap) * 32 + co) * 32 + ci];
      const double want = (co < Cout && ci < Cin) ? ref[(tap * Cout + co) * Cin + ci] : 0.0;
      maxerr = fmax(maxerr, fabs(got - want)); maxref = fmax(maxref, fabs(want));
    }
    printf("rc %d  max abs err %.3e  (max |ref| %.3e)  %s\n", rc, maxerr, maxref, maxerr <= 1e-3 * maxref ? "OK" : "MISMATCH");
    hipFree(x); hipFree(dy); hipFree(part);
  }
  {  // timing at the stage-0 shape
    const int N = 2, D = 160, H = 160, W = 256, Cin = 24, Cout = 24;
    const long V = (long)N * D * H * W;
    unsigned short *x, *dy; float* part;
    hipMalloc(&x, V * Cin * 2); hipMalloc(&dy, V * Cout * 2);
    hipMemset(x, 0x3c, V * Cin * 2); hipMemset(dy, 0x3c, V * Cout * 2);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int n_wg : {256, 512, 768, 1024, 1536}) {
      hipMalloc(&part, (size_t)n_wg * 27 * 1024 * 4);
      for (int i = 0; i < 2; ++i) transoar_conv3d_k3_wgrad_lds(x, dy, part, n_wg, N, D, H, W, Cin, Cout, 0, Cin, 0, Cout, nullptr);
      hipEventRecord(a);
      for (int i = 0; i < 5; ++i) transoar_conv3d_k3_wgrad_lds(x, dy, part, n_wg, N, D, H, W, Cin, Cout, 0, Cin, 0, Cout, nullptr);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("n_wg %4d: %.3f ms  (%s)\n", n_wg, ms / 5, hipGetErrorString(hipGetLastError()));
      hipFree(part);
    }
  }
  return 0;
}
