/*
 * transoar_instnorm.h -- C ABI of the fused InstanceNorm3d(affine) + ReLU
 * kernels for gfx950 (bf16 channels-last activations, fp32/fp64 statistics).
 *
 * Replaces the norm + activation pair of the reference's encoder blocks
 *   nn.InstanceNorm3d(num_features, affine=True, eps=1e-5) ; nn.ReLU(inplace)
 *   transoar/models/backbones/encoder_blocks.py:34-36 and :44-46
 * (forward and backward).  The Python autograd shim is transoar_amd/instnorm.py.
 *
 *   x, y, dy, dx  (N, V, C) bf16, V = D*H*W voxels (channels-last), C % 8 == 0
 *                 and 192 % (C/8) == 0 (every backbone width 24..768 qualifies)
 *   gamma, beta   (C) fp32
 *   mean_rstd     (N, C, 2) fp32: written by forward, read by backward
 *   stats_ws / red_ws  (N, C, 2) fp64 scratch
 * All pointers are device pointers, 16-byte aligned; calls are asynchronous on
 * `hip_stream`.  Returns 0, a hipError_t (> 0), or a negative code below.
 */
#ifndef TRANSOAR_INSTNORM_H
#define TRANSOAR_INSTNORM_H

#ifdef __cplusplus
extern "C" {
#endif

enum {
  TRANSOAR_IN_OK = 0,
  TRANSOAR_IN_ERR_NULL = -1,
  TRANSOAR_IN_ERR_DIM = -2,
  TRANSOAR_IN_ERR_CHANNELS = -3
};

/* y = relu?( (x - mean_nc) / sqrt(var_nc + eps) * gamma_c + beta_c ), biased variance */
int transoar_instnorm_relu_forward(const void* x, const float* gamma, const float* beta, void* y,
                                   double* stats_ws, float* mean_rstd, int N, long V, int C,
                                   float eps, int relu, void* hip_stream);

/* The same with the statistics already taken in the epilogue of the producing convolution (round 4;
 * transoar_conv3d_k3_forward_stats in transoar_conv3d.h): stat_part (N * rows_per_sample, 2, 32) fp32 holds per row of
 * conv tiles the sums and sums of squares of the bf16 outputs; a small kernel adds them in fp64 into stats_ws, then
 * the apply pass of the entry above runs.  C <= 32.  (encoder_blocks.py:28-36: Conv3d -> InstanceNorm3d -> ReLU) */
int transoar_instnorm_relu_forward_parts(const void* x, const float* gamma, const float* beta, void* y,
                                         const float* stat_part, int rows_per_sample, double* stats_ws,
                                         float* mean_rstd, int N, long V, int C, float eps, int relu,
                                         void* hip_stream);

/* dx of the above; red_ws ends up holding, per (n,c), {sum g, sum g*xhat} with
 * g = dy * [relu active]: dbeta_c = sum_n red[n][c][0], dgamma_c = sum_n red[n][c][1];
 * dparams (2, C) fp32, optional (NULL: not written): row 0 = dbeta, row 1 = dgamma (ABI 3). */
int transoar_instnorm_relu_backward(const void* x, const void* dy, const float* gamma,
                                    const float* beta, const float* mean_rstd, void* dx,
                                    double* red_ws, float* dparams, int N, long V, int C, int relu,
                                    void* hip_stream);

int transoar_instnorm_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TRANSOAR_INSTNORM_H */
