/*
 * TEST INFRASTRUCTURE ONLY -- never linked, imported or called by the product
 * path (transoar_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker.
 *
 * Scalar CPU restatement of the reference's 3-D multi-scale deformable
 * attention operator.  This header is included once per REAL type
 * (float, double) by msda3d_oracle.c.  It restates, in sequential C:
 *
 *   forward   transoar/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:370-439
 *             (ms_deformable_im2col_gpu_kernel) with the 8-corner sampler of
 *             :31-114 (ms_deform_attn_im2col_trilinear)
 *   backward  :116-241 (ms_deform_attn_col2im_trilinear) driven the way the
 *             shared-memory kernels do (:441-661): per (b,q,m,l,p) the channel
 *             partials of grad_attn / grad_loc are summed over c, grad_value is
 *             a scatter-add.
 *
 * Semantics pinned:
 *   value (N,S,M,C) channel-last, level l occupies rows [lsi[l], lsi[l]+D*H*W),
 *   voxel (d,h,w) is row (d*H+h)*W+w; loc is (x,y,z)=(w,h,d) in [0,1];
 *   pixel coordinate = loc*size - 0.5 (== grid_sample align_corners=False);
 *   a point contributes iff -1 < coord < size on all three axes (strict);
 *   every corner is bounds-checked on its own -> zero padding.
 * All arithmetic is done in REAL so the float build rounds like a float kernel.
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* corner k in 0..7: bit2 = d-high, bit1 = h-high, bit0 = w-high.  This is the
 * v1..v8 order of the reference (.cuh:60-107): v1=(lo,lo,lo), v2=(lo,lo,hi)... */

void FN(msda3d_oracle_forward_)(const REAL* value, const int64_t* shapes,
                                const int64_t* lsi, const REAL* loc,
                                const REAL* attn, REAL* out, int N, int S,
                                int M, int C, int L, int Lq, int P) {
  const long row = (long)M * C;
  for (long b = 0; b < N; ++b)
    for (long q = 0; q < Lq; ++q)
      for (long m = 0; m < M; ++m) {
        const long bqm = (b * Lq + q) * M + m;
        REAL* o = out + bqm * C;
        for (int c = 0; c < C; ++c) o[c] = (REAL)0;
        for (int l = 0; l < L; ++l) {
          const int D = (int)shapes[3 * l], H = (int)shapes[3 * l + 1],
                    W = (int)shapes[3 * l + 2];
          const REAL* base = value + (b * S + lsi[l]) * row + m * C;
          for (int p = 0; p < P; ++p) {
            const long lp = (bqm * L + l) * P + p;
            const REAL a = attn[lp];
            const REAL w_im = loc[3 * lp + 0] * W - (REAL)0.5;
            const REAL h_im = loc[3 * lp + 1] * H - (REAL)0.5;
            const REAL d_im = loc[3 * lp + 2] * D - (REAL)0.5;
            if (!(d_im > -1 && h_im > -1 && w_im > -1 && d_im < D &&
                  h_im < H && w_im < W))
              continue;
            const int d0 = (int)FLOOR(d_im), h0 = (int)FLOOR(h_im),
                      w0 = (int)FLOOR(w_im);
            const REAL ld = d_im - d0, lh = h_im - h0, lw = w_im - w0;
            const REAL fd[2] = {1 - ld, ld}, fh[2] = {1 - lh, lh},
                       fw[2] = {1 - lw, lw};
            for (int c = 0; c < C; ++c) {
              /* the reference forms val = w1*v1 + ... + w8*v8 left to right,
               * then col += val * weight (.cuh:112, :430) */
              REAL val = 0;
              for (int k = 0; k < 8; ++k) {
                const int dd = k >> 2, dh = (k >> 1) & 1, dw = k & 1;
                const int d = d0 + dd, h = h0 + dh, w = w0 + dw;
                REAL v = 0;
                if (d >= 0 && d < D && h >= 0 && h < H && w >= 0 && w < W)
                  v = base[(((long)d * H + h) * W + w) * row + c];
                val += (fd[dd] * fh[dh] * fw[dw]) * v;
              }
              o[c] += val * a;
            }
          }
        }
      }
}

/* No pre-zeroed outputs are assumed: this routine zeroes all three itself (reference: at::zeros_like,
 * ms_deform_attn_cuda.cu:122-124). */
void FN(msda3d_oracle_backward_)(const REAL* value, const int64_t* shapes,
                                 const int64_t* lsi, const REAL* loc,
                                 const REAL* attn, const REAL* grad_out,
                                 REAL* grad_value, REAL* grad_loc,
                                 REAL* grad_attn, int N, int S, int M, int C,
                                 int L, int Lq, int P) {
  const long row = (long)M * C;
  memset(grad_value, 0, sizeof(REAL) * (size_t)N * S * row);
  memset(grad_loc, 0, sizeof(REAL) * (size_t)N * Lq * M * L * P * 3);
  memset(grad_attn, 0, sizeof(REAL) * (size_t)N * Lq * M * L * P);
  for (long b = 0; b < N; ++b)
    for (long q = 0; q < Lq; ++q)
      for (long m = 0; m < M; ++m) {
        const long bqm = (b * Lq + q) * M + m;
        const REAL* g = grad_out + bqm * C;
        for (int l = 0; l < L; ++l) {
          const int D = (int)shapes[3 * l], H = (int)shapes[3 * l + 1],
                    W = (int)shapes[3 * l + 2];
          const long off = (b * S + lsi[l]) * row + m * C;
          const REAL* base = value + off;
          REAL* gbase = grad_value + off;
          for (int p = 0; p < P; ++p) {
            const long lp = (bqm * L + l) * P + p;
            const REAL a = attn[lp];
            const REAL w_im = loc[3 * lp + 0] * W - (REAL)0.5;
            const REAL h_im = loc[3 * lp + 1] * H - (REAL)0.5;
            const REAL d_im = loc[3 * lp + 2] * D - (REAL)0.5;
            if (!(d_im > -1 && h_im > -1 && w_im > -1 && d_im < D &&
                  h_im < H && w_im < W))
              continue; /* grads of a skipped point stay zero (.cuh:598-606) */
            const int d0 = (int)FLOOR(d_im), h0 = (int)FLOOR(h_im),
                      w0 = (int)FLOOR(w_im);
            const REAL ld = d_im - d0, lh = h_im - h0, lw = w_im - w0;
            const REAL fd[2] = {1 - ld, ld}, fh[2] = {1 - lh, lh},
                       fw[2] = {1 - lw, lw};
            REAL ga = 0, gw = 0, gh = 0, gd = 0;
            for (int c = 0; c < C; ++c) {
              const REAL top = g[c];
              const REAL top_a = top * a; /* top_grad_value, .cuh:151 */
              REAL val = 0, dwt = 0, dht = 0, ddt = 0;
              for (int k = 0; k < 8; ++k) {
                const int dd = k >> 2, dh = (k >> 1) & 1, dw = k & 1;
                const int d = d0 + dd, h = h0 + dh, w = w0 + dw;
                if (!(d >= 0 && d < D && h >= 0 && h < H && w >= 0 && w < W))
                  continue;
                const long idx = (((long)d * H + h) * W + w) * row + c;
                const REAL v = base[idx];
                const REAL wt = fd[dd] * fh[dh] * fw[dw];
                /* derivative of the corner weight along one axis is +-(product
                 * of the other two axis weights): - for the low corner, + for
                 * the high corner (.cuh:159-231) */
                ddt += (dd ? (REAL)1 : (REAL)-1) * fh[dh] * fw[dw] * v;
                dht += (dh ? (REAL)1 : (REAL)-1) * fd[dd] * fw[dw] * v;
                dwt += (dw ? (REAL)1 : (REAL)-1) * fd[dd] * fh[dh] * v;
                val += wt * v;
                gbase[idx] += wt * top_a;
              }
              ga += top * val;        /* .cuh:236 */
              gw += W * dwt * top_a;  /* .cuh:240 */
              gh += H * dht * top_a;  /* .cuh:239 */
              gd += D * ddt * top_a;  /* .cuh:238 */
            }
            grad_attn[lp] = ga;
            grad_loc[3 * lp + 0] = gw;
            grad_loc[3 * lp + 1] = gh;
            grad_loc[3 * lp + 2] = gd;
          }
        }
      }
}

#undef FN
#undef CAT
#undef CAT_
