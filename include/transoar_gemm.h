/*
 * transoar_gemm.h -- C ABI of the gfx950 bf16/f16 token GEMM  C = A . B^T (+ bias) (ReLU).
 *
 * Replaces, on the refinement block's 234 000-token projections, what the reference reaches through
 * nn.Linear -> cuBLAS under autocast: ops/modules/ms_deform_attn.py:109-140 (value_proj, sampling_offsets,
 * attention_weights, output_proj) and backbones/decoder_blocks.py:157-174 (linear1 + ReLU, linear2), forward
 * and data gradient (dX = dY . W is the same NT product with W^T as the B operand).
 *
 *   A    (M, K) row-major, leading dimension lda (elements)     activations / output gradients
 *   B    (N, K) row-major, leading dimension ldb                nn.Linear's weight layout
 *   bias (N,) fp32 or NULL
 *   C    (M, N) row-major, leading dimension ldc; out_dtype = in_dtype (bf16/f16) or TRANSOAR_GEMM_F32
 * K a multiple of 8, N a multiple of 4, lda/ldb multiples of 8, ldc of 4; buffers 16-byte aligned; fp32
 * accumulation.  Asynchronous on `hip_stream`.  Returns 0, a negative argument error, or a hipError_t.
 */
#ifndef TRANSOAR_GEMM_H
#define TRANSOAR_GEMM_H
#ifdef __cplusplus
extern "C" {
#endif

enum { TRANSOAR_GEMM_F32 = 0, TRANSOAR_GEMM_BF16 = 2, TRANSOAR_GEMM_F16 = 3 };
enum {
  TRANSOAR_GEMM_ERR_NULL = -1,
  TRANSOAR_GEMM_ERR_DIM = -2,
  TRANSOAR_GEMM_ERR_DTYPE = -3,
  TRANSOAR_GEMM_ERR_ALIGN = -4
};

int transoar_gemm_nt(const void* A, const void* B, const float* bias, void* C, int M, int N, int K, int lda, int ldb,
                     int ldc, int in_dtype, int out_dtype, int relu, void* hip_stream);
/*
 * transoar_gemm_nt (bf16 in, bf16 out) with the exact GELU of the Swin blocks' MLP in its epilogue (reference:
 * backbones/encoder_blocks.py Mlp: fc2(act(fc1(x))), act = nn.GELU; forward and backward of `act` fused into the products
 * on either side of it).  `aux` (M, N) bf16 has C's leading dimension.
 *   TRANSOAR_GEMM_GELU_FORWARD:   aux = h = A B^T + bias (written),  C = gelu(h)            -- fc1 + act
 *   TRANSOAR_GEMM_GELU_BACKWARD:  C = (A B^T) * gelu'(aux)           (aux = h, read)        -- act's backward on fc2's data gradient
 * Both apply the activation to the bf16-rounded product, in fp32 (Phi(x) to 1.5e-7 absolute): the rounding points of the
 * separate kernels they replace.
 */
enum { TRANSOAR_GEMM_GELU_FORWARD = 1, TRANSOAR_GEMM_GELU_BACKWARD = 2 };
int transoar_gemm_nt_gelu(const void* A, const void* B, const float* bias, void* C, void* aux, int M, int N, int K, int lda,
                          int ldb, int ldc, int mode, void* hip_stream);
/*
 * The same product for the two shape classes of the refinement block, dense operands (leading dimension = row length),
 * bf16 in and out, register-stationary / tile-streaming kernels (csrc/gemm_stream.hip):
 *   transoar_gemm_k384:  C (M, N) = A (M, 384) . B (N, 384)^T (+ bias) (ReLU);  N a multiple of 64
 *   transoar_gemm_n384:  C (M, 384) = A (M, K) . B (384, K)^T (+ bias);          K a multiple of 32
 */
int transoar_gemm_k384(const void* A, const void* B, const float* bias, void* C, int M, int N, int relu, void* hip_stream);
int transoar_gemm_n384(const void* A, const void* B, const float* bias, void* C, int M, int K, void* hip_stream);
/* transoar_gemm_k384 with seeded dropout in the epilogue: y = keep ? act(x W^T + b) * keep_scale : 0, the mask of
 * csrc/tokens.hip's relu_dropout (element pair p kept where the 16-bit halves of hash32(p * 0x9e3779b9 + *drop_seed) are
 * below keep_prob * 2^16): decoder_blocks.py:166's dropout(activation(linear1(x))) in one kernel.  drop_seed: device int. */
int transoar_gemm_k384_drop(const void* A, const void* B, const float* bias, void* C, int M, int N, int relu,
                            const int* drop_seed, float keep_prob, float keep_scale, void* hip_stream);

/* Weight gradient of a token linear layer with 384 channels on one side (reference: the autograd of nn.Linear in
 * ops/modules/ms_deform_attn.py:109-140 and backbones/decoder_blocks.py:157-174), fp32 out:
 *   out = A^T B,  A (T, Na) bf16, B (T, 384) bf16, Na a multiple of 128;
 *   transpose_out == 0: out (Na, 384) row-major;  != 0: out (384, Na) row-major (the case "384 is the layer's OUTPUT width").
 * The tokens are split into `chunks` = transoar_gemm_wgrad384_chunks(T, Na) ranges, one workgroup per range and column
 * tile; `part` holds chunks * Na * 384 floats of partial sums (device scratch), summed into `out` by a second kernel. */
/* transoar_gemm_wgrad384_bias additionally sums one operand over the tokens (the layer's bias gradient, sum_t dY[t][:]),
 * from the fragments the product has in registers anyway: bias_side 1 = A's Na columns, 2 = B's 384 columns, 0 = none;
 * bias_part: chunks * (Na or 384) floats of scratch, bias_out: the sums (fp32). */
int transoar_gemm_wgrad384_bias(const void* A, const void* B, float* part, float* out, int T, int Na, int transpose_out,
                                int chunks, float* bias_part, float* bias_out, int bias_side, void* hip_stream);
int transoar_gemm_wgrad384_chunks(int T, int Na);
int transoar_gemm_wgrad384(const void* A, const void* B, float* part, float* out, int T, int Na, int transpose_out,
                           int chunks, void* hip_stream);

/* The data gradient of the FFN's second layer with the gradient of dropout(relu(.)) in its epilogue
 * (decoder_blocks.py:166-167 backwards): C (M, N) = gate > 0 ? (A (M, 384) . B (N, 384)^T) * scale : 0, gate (M, N) bf16 =
 * the hidden tensor the forward saved (positive exactly where kept and active), scale = 1 / keep_prob. */
int transoar_gemm_k384_gate(const void* A, const void* B, const void* gate, void* C, int M, int N, float scale, void* hip_stream);

int transoar_gemm_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TRANSOAR_GEMM_H */
