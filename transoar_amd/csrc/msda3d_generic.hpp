// Generic (any C, any dtype incl. fp64) MSDeformAttn-3D kernels: one wave per
// (batch, query, head) item, lanes stride over the channels, grad_value through
// hardware fp atomics.  Correctness path for the reference's gradcheck channel
// sweep (ops/test.py:122) and for head sizes the vector kernels do not cover;
// not a performance path.
#pragma once
#include "msda3d_common.hpp"

namespace transoar {

// ---------------------------------------------------------------------------
// generic kernels: any C, any dtype.  One wave per item, lanes stride over the
// channels.  Used for the reference's odd gradcheck channel counts
// (ops/test.py:122) and whenever C*elt is not 128/256/512 bytes.
// ---------------------------------------------------------------------------
template <typename VT, typename LT>
__global__ __launch_bounds__(64 * kWavesPerBlock) void msda3d_fwd_generic(
    const VT* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const LT* __restrict__ loc,
    const LT* __restrict__ attn, VT* __restrict__ out, int S, int M, int C,
    int L, int Lq, int P, long n_items) {
  using A = typename Elem<VT>::acc;
  const int lane = threadIdx.x & 63;
  const long item = static_cast<long>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (item >= n_items) return;
  const int m = static_cast<int>(item % M);
  const long b = (item / M) / Lq;
  const long row_stride = static_cast<long>(M) * C;
  const VT* vhead = value + (b * S * M + m) * C;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane;
    A acc = A(0);
    for (int l = 0; l < L; ++l) {
      const int D = static_cast<int>(shapes[3 * l]), H = static_cast<int>(shapes[3 * l + 1]),
                W = static_cast<int>(shapes[3 * l + 2]);
      const long start = lsi[l];
      for (int p = 0; p < P; ++p) {
        const long j = (item * L + l) * P + p;
        const A a = static_cast<A>(Elem<LT>::ld(attn + j));
        const A w_im = pixel_coord(static_cast<A>(Elem<LT>::ld(loc + 3 * j)), W);
        const A h_im = pixel_coord(static_cast<A>(Elem<LT>::ld(loc + 3 * j + 1)), H);
        const A d_im = pixel_coord(static_cast<A>(Elem<LT>::ld(loc + 3 * j + 2)), D);
        if (!(d_im > A(-1) && h_im > A(-1) && w_im > A(-1) && d_im < D && h_im < H && w_im < W))
          continue;
        const A fd = floor(d_im), fh = floor(h_im), fw = floor(w_im);
        const int d0 = static_cast<int>(fd), h0 = static_cast<int>(fh), w0 = static_cast<int>(fw);
        const A ld = d_im - fd, lh = h_im - fh, lw = w_im - fw;
        A val = A(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int dd = k >> 2, dh = (k >> 1) & 1, dw = k & 1;
          const int d = d0 + dd, h = h0 + dh, w = w0 + dw;
          if (c < C && static_cast<unsigned>(d) < static_cast<unsigned>(D) &&
              static_cast<unsigned>(h) < static_cast<unsigned>(H) &&
              static_cast<unsigned>(w) < static_cast<unsigned>(W)) {
            const long r = start + (static_cast<long>(d) * H + h) * W + w;
            const A wt = (dd ? ld : A(1) - ld) * (dh ? lh : A(1) - lh) * (dw ? lw : A(1) - lw);
            val += wt * static_cast<A>(Elem<VT>::ld(vhead + r * row_stride + c));
          }
        }
        acc += val * a;
      }
    }
    if (c < C) Elem<VT>::st(out + item * C + c, acc);
  }
}

template <typename VT, typename LT>
__global__ __launch_bounds__(64 * kWavesPerBlock) void msda3d_bwd_generic(
    const VT* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const LT* __restrict__ loc,
    const LT* __restrict__ attn, const VT* __restrict__ grad_out,
    typename Elem<VT>::acc* __restrict__ grad_value, LT* __restrict__ grad_loc,
    LT* __restrict__ grad_attn, int S, int M, int C, int L, int Lq, int P,
    long n_items) {
  using A = typename Elem<VT>::acc;
  const int lane = threadIdx.x & 63;
  const long item = static_cast<long>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (item >= n_items) return;
  const int m = static_cast<int>(item % M);
  const long b = (item / M) / Lq;
  const long row_stride = static_cast<long>(M) * C;
  const long head_off = (b * S * M + m) * C;
  const VT* vhead = value + head_off;
  A* gvhead = grad_value + head_off;
  for (int l = 0; l < L; ++l) {
    const int D = static_cast<int>(shapes[3 * l]), H = static_cast<int>(shapes[3 * l + 1]),
              W = static_cast<int>(shapes[3 * l + 2]);
    const long start = lsi[l];
    for (int p = 0; p < P; ++p) {
      const long j = (item * L + l) * P + p;
      const A a = static_cast<A>(Elem<LT>::ld(attn + j));
      const A w_im = pixel_coord(static_cast<A>(Elem<LT>::ld(loc + 3 * j)), W);
      const A h_im = pixel_coord(static_cast<A>(Elem<LT>::ld(loc + 3 * j + 1)), H);
      const A d_im = pixel_coord(static_cast<A>(Elem<LT>::ld(loc + 3 * j + 2)), D);
      A pa = A(0), pw = A(0), ph = A(0), pd = A(0);
      if (d_im > A(-1) && h_im > A(-1) && w_im > A(-1) && d_im < D && h_im < H && w_im < W) {
        const A fd = floor(d_im), fh = floor(h_im), fw = floor(w_im);
        const int d0 = static_cast<int>(fd), h0 = static_cast<int>(fh), w0 = static_cast<int>(fw);
        const A ld = d_im - fd, lh = h_im - fh, lw = w_im - fw;
        for (int c = lane; c < C; c += 64) {
          const A top = static_cast<A>(Elem<VT>::ld(grad_out + item * C + c));
          const A top_a = top * a;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int dd = k >> 2, dh = (k >> 1) & 1, dw = k & 1;
            const int d = d0 + dd, h = h0 + dh, w = w0 + dw;
            if (static_cast<unsigned>(d) < static_cast<unsigned>(D) &&
                static_cast<unsigned>(h) < static_cast<unsigned>(H) &&
                static_cast<unsigned>(w) < static_cast<unsigned>(W)) {
              const A wd = dd ? ld : A(1) - ld, wh = dh ? lh : A(1) - lh, ww = dw ? lw : A(1) - lw;
              const long r = start + (static_cast<long>(d) * H + h) * W + w;
              const A v = static_cast<A>(Elem<VT>::ld(vhead + r * row_stride + c));
              atomic_accum(gvhead + r * row_stride + c, wd * wh * ww * top_a);
              pa += wd * wh * ww * v * top;
              pw += (dw ? A(1) : A(-1)) * wd * wh * v * top_a;
              ph += (dh ? A(1) : A(-1)) * wd * ww * v * top_a;
              pd += (dd ? A(1) : A(-1)) * wh * ww * v * top_a;
            }
          }
        }
      }
#pragma unroll
      for (int mask = 1; mask < 64; mask <<= 1) {
        pa += xor_lanes(pa, mask);
        pw += xor_lanes(pw, mask);
        ph += xor_lanes(ph, mask);
        pd += xor_lanes(pd, mask);
      }
      if (lane == 0) {
        Elem<LT>::st(grad_attn + j, pa);
        Elem<LT>::st(grad_loc + 3 * j, pw * W);
        Elem<LT>::st(grad_loc + 3 * j + 1, ph * H);
        Elem<LT>::st(grad_loc + 3 * j + 2, pd * D);
      }
    }
  }
}

}  // namespace transoar
