/*
 * transoar_conv3d.h -- C ABI of the gfx950 3x3x3 convolution kernels of the
 * AttnFPN backbone (bf16 storage, fp32 accumulation on the matrix cores,
 * channels-last NDHWC activations).
 *
 * These entry points replace the cuDNN convolutions PyTorch runs for
 *   nn.Conv3d(k=3, pad=1, stride 1|2, bias=False)   encoder stages
 *       transoar/models/backbones/encoder_blocks.py:28-48
 *   nn.Conv3d(k=3, pad=1, bias=True)                 FPN output convs
 *       transoar/models/backbones/attn_fpn.py:65-73, :126
 * (forward, data gradient and weight gradient).  The Python module
 * transoar_amd/conv3d.py is the autograd shim that binds them.
 *
 * All pointers are device pointers, 16-byte aligned; every call is
 * asynchronous on `hip_stream`.  Returns 0, a hipError_t (> 0) or one of the
 * negative codes below.
 */
#ifndef TRANSOAR_CONV3D_H
#define TRANSOAR_CONV3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  TRANSOAR_CONV_OK = 0,
  TRANSOAR_CONV_ERR_NULL = -1,
  TRANSOAR_CONV_ERR_DIM = -2,       /* bad size, or a tensor >= 4 GiB (32-bit offsets) */
  TRANSOAR_CONV_ERR_CHANNELS = -3   /* Cin % 8 or Cout % 4 (k3), Cout % 8 (c1)        */
};

/*
 * y = conv3d(x, w, pad 1, stride) (+ bias), implicit GEMM.
 *   x    (N, D, H, W, Cin)        bf16
 *   wk   (27, Cout, Cin)          bf16, tap index = (kd*3 + kh)*3 + kw
 *   bias (Cout) fp32 or NULL
 *   y    (N, Do, Ho, Wo, Cout)    bf16, Do = (D-1)/stride + 1 ...
 * dilated_input != 0 (stride must be 1): x is read as if zero-dilated by 2
 * (only even coordinates exist) and the output grid is (2D, 2H, 2W): with
 * flipped, in/out-swapped weights this is the data gradient of a stride-2
 * layer.  The data gradient of a stride-1 layer is a plain call with those
 * weights.
 */
int transoar_conv3d_k3_forward(const void* x, const void* wk, const float* bias, void* y,
                               int N, int D, int H, int W, int Cin, int Cout, int stride,
                               int dilated_input, void* hip_stream);

/*
 * dW (27, Cout, Cin) fp32 += sum over voxels  dy[voxel][cout] * x[voxel+tap][cin]
 * (dw must be zeroed by the caller; accumulated with fp32 atomics).
 *   gyT  (Cout, N, Do, Ho, Wo)        bf16, channels FIRST
 *   xT3  (3, Cin, N, D, H, Wo)        bf16, channels first, three copies pre-
 *        shifted (and for stride 2 decimated) along W:
 *        xT3[kw][ci][n][d][h][wo] = x[n][d][h][stride*wo + kw - 1][ci], 0 outside
 *   W here is the row length of xT3 and must equal Wo; Wo % 8 == 0.
 *   stride_dh is the layer's stride, applied along D and H.
 */
int transoar_conv3d_k3_wgrad(const void* gyT, const void* xT3, float* dw, int N, int D, int H,
                             int W, int Cin, int Do, int Ho, int Wo, int Cout, int stride_dh,
                             void* hip_stream);

/* The Cin = 1 stem on the matrix cores without a channel-padded copy of the volume: x (N, D, H, W) bf16,
 * wk (27, Cout, 8) bf16 with the real input channel first, y (N, D, H, W, Cout) bf16; stride 1, Cout % 4 == 0, <= 32
 * (encoder_blocks.py:28-35 with in_channels = 1). */
int transoar_conv3d_k3_forward_c1(const void* x, const void* wk, const float* bias, void* y, int N, int D, int H,
                                  int W, int Cout, void* hip_stream);

/* Forward of a full-resolution stride-1 layer (Cin = 1 | 8 | 16 | 24, Cout <= 32: the halo-tile kernel) that ALSO leaves
 * the InstanceNorm statistics of its output behind: stat_part (rows, 2, 32) fp32 = per row of tiles (rows =
 * transoar_conv3d_k3_stat_rows(...), 0 = the layer is not covered) the per-channel sum and sum of squares of the bf16
 * outputs.  Cin == 1: x as for transoar_conv3d_k3_forward_c1 (wk (27, Cout, 8)), else as transoar_conv3d_k3_forward. */
int transoar_conv3d_k3_stat_rows(int N, int D, int H, int W, int Cin, int Cout, int stride);
int transoar_conv3d_k3_forward_stats(const void* x, const void* wk, const float* bias, void* y, float* stat_part,
                                     int N, int D, int H, int W, int Cin, int Cout, void* hip_stream);

/*
 * First layer, Cin == 1, stride 1: stencil.
 *   x (N, D, H, W) bf16 ; w (27, Cout) fp32 ; y (N, D, H, W, Cout) bf16 ; Cout % 8 == 0, Cout <= 64
 */
int transoar_conv3d_c1_forward(const void* x, const float* w, void* y, int N, int D, int H,
                               int W, int Cout, void* hip_stream);

/*
 * Weight gradient, stride 1 / pad 1, of one block of at most 32 x 32 channels; W % 64 == 0.
 *   x (N, D, H, W, Cin) bf16 ; dy (N, D, H, W, Cout) bf16 (both NDHWC), Cin and Cout multiples of 8
 *   block = input channels [ci0, ci0 + ci_n), output channels [co0, co0 + co_n); ci_n, co_n multiples
 *   of 8 and <= 32 (a 48 -> 48 layer takes 4 calls)
 *   partial (n_wg, 27, 32, 32) fp32, written completely: partial[g][tap][cout - co0][cin - ci0];
 *   dW[cout][cin][tap] = sum_g partial[g][tap][cout - co0][cin - ci0]   (tap = (kd*3+kh)*3+kw).
 * n_wg workgroups (two per CU) share the N*D*H*(W/64) row segments.
 */
int transoar_conv3d_k3_wgrad_lds(const void* x, const void* dy, float* partial, int n_wg, int N, int D,
                                 int H, int W, int Cin, int Cout, int ci0, int ci_n, int co0, int co_n,
                                 void* hip_stream);
/*
 * Second version of the same (same arguments, same partial layout): tiles staged as they lie in memory and transposed
 * by the LDS read (ds_read_b64_tr_b16), four waves per workgroup sharing the 27 taps 7/7/7/6.  n_wg: 3 per CU.
 */
int transoar_conv3d_k3_wgrad_tr(const void* x, const void* dy, float* partial, int n_wg, int N, int D,
                                 int H, int W, int Cin, int Cout, int ci0, int ci_n, int co0, int co_n,
                                 void* hip_stream);

/*
 * Weight gradient of the Cin == 1 first layer (stride 1, pad 1):
 *   x (N, D, H, W) bf16 ; dy (N, D, H, W, Cout) bf16 ; Cout <= 32 ; W % 16 == 0
 *   partial (n_partial, 32, 32) fp32, written completely: partial[p][cout][tap] (tap = (kd*3+kh)*3+kw,
 *   entries with cout >= Cout or tap >= 27 are zero); dW[cout][tap] = sum_p partial[p][cout][tap].
 * n_partial workgroups share the N*D*H rows of the volume (a few thousand fill the chip).
 */
int transoar_conv3d_c1_wgrad(const void* x, const void* dy, float* partial, int n_partial, int N, int D,
                             int H, int W, int Cout, void* hip_stream);
/*
 * The same with coalesced 16-byte loads (dy rows through LDS and the transposing LDS read, x rows staged three times,
 * shifted by kw): Cout % 8 == 0, W % 64 == 0, W <= 256.  n_partial: about 4 workgroups per CU; each walks
 * ceil(N*D*H / n_partial) consecutive W-rows.
 */
int transoar_conv3d_c1_wgrad_tr(const void* x, const void* dy, float* partial, int n_partial, int N, int D,
                                int H, int W, int Cout, void* hip_stream);

/*
 * bf16 layout change between channels-last (N, V, C) and channels-first
 * (N, C, V), V = D*H*W, C % 8 == 0.  to_channels_first != 0: in is (N,V,C).
 * Used where a hand-written (NDHWC) layer meets a torch/MIOpen (NCDHW) one.
 */
int transoar_layout_bf16(const void* in, void* out, int N, long V, int C, int to_channels_first,
                         void* hip_stream);

int transoar_conv3d_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TRANSOAR_CONV3D_H */
