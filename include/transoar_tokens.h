/*
 * transoar_tokens.h -- C ABI of the fused token-stream kernels of the FPN
 * "refine" block for gfx950: residual add + LayerNorm + the casts and the
 * positional add that surround it, one pass over the (N*S, C) token matrix
 * instead of six.
 *
 * Replaces, per DefAttnLayer (transoar/models/backbones/decoder_blocks.py:143-177):
 *     src = norm(src + dropout(branch))                  :167-168 / :172-174
 * and the head of the next layer / MSDeformAttn input casts under autocast
 *     query = src + pos  (with_pos_embed, :157-158, :164), pos = sine + level_embed[l] (:76-83)
 *
 *   x        (rows, cols) residual stream, fp32 or bf16      rows = N*S tokens
 *   r        (rows, cols) bf16 branch output (after dropout), or NULL
 *   y32      (rows, cols) fp32   LayerNorm output (what autocast's layer_norm returns)
 *   y16      (rows, cols) bf16   the same, rounded: what the next nn.Linear reads
 *   q16      (rows, cols) bf16   round(y32 + (pos_sine[s] + level_embed[l(s)])), or NULL
 *   pos_sine (S, cols) fp32, shared by the batch; level_embed (L, cols) fp32;
 *   level_start (L) int32 first token of each level, S = tokens per batch element
 *   keep     (rows, cols) mask bytes of the branch's dropout (r is the branch BEFORE dropout:
 *            the kernels apply keep ? r * keep_scale : 0), or NULL when r is already final
 *   keep_seed, keep_prob: the dropout mask without mask bytes (keep == NULL): element pair p = e / 2 of the
 *            tensor is kept where 16-bit halves of hash32(p * 0x9e3779b9 + *keep_seed) are below
 *            keep_prob * 65536 (hash32: the two-multiply xorshift finaliser in csrc/tokens.hip); *keep_seed is
 *            one int32 on the device, drawn per call by the host from its generator.  NULL = no such mask.
 *   mean_rstd (rows, 2) fp32 written by forward, read by backward
 * cols must be a multiple of 128 and <= 1024.  Device pointers, 16-byte
 * aligned, asynchronous on `hip_stream`.  Returns 0, a hipError_t (> 0) or a
 * negative code.
 */
#ifndef TRANSOAR_TOKENS_H
#define TRANSOAR_TOKENS_H

#ifdef __cplusplus
extern "C" {
#endif

enum {
  TRANSOAR_TOK_OK = 0,
  TRANSOAR_TOK_ERR_NULL = -1,
  TRANSOAR_TOK_ERR_DIM = -2,
  TRANSOAR_TOK_ERR_LEVELS = -3
};
#define TRANSOAR_TOK_MAX_LEVELS 8

int transoar_add_layernorm_forward(const void* x, int x_is_bf16, const void* r, const float* weight,
                                   const float* bias, float eps, const float* pos_sine,
                                   const float* level_embed, const int* level_start, int L, long S,
                                   float* y32, void* y16, void* q16, float* mean_rstd, long rows,
                                   int cols, const unsigned char* keep, float keep_scale,
                                   const int* keep_seed, float keep_prob, void* hip_stream);

/*
 * Backward.  g32 / g16 / gq16 are the gradients w.r.t. y32 / y16 / q16 (any of
 * them may be NULL = zero).  Writes gx (dtype of x) and, when gr16 is given, the
 * branch gradient gr16 = bf16(keep ? gx * keep_scale : 0)  (without a mask and with a
 * bf16 x the caller may use gx for both and pass NULL).
 * partials: (transoar_add_layernorm_partial_rows(), (2 + L) * cols) fp32, written
 * completely: per persistent wave the column sums of g*xhat (-> d weight), g (-> d bias)
 * and, per level, of gq (-> d level_embed); the caller sums over dim 0.
 */
int transoar_add_layernorm_backward(const float* g32, const void* g16, const void* gq16, const void* x,
                                    int x_is_bf16, const void* r, const float* weight,
                                    const float* mean_rstd, const int* level_start, int L, long S,
                                    void* gx, void* gr16, float* partials, long rows, int cols,
                                    const unsigned char* keep, float keep_scale, const int* keep_seed,
                                    float keep_prob, void* hip_stream);

/*
 * FFN activation of the layer, one pass each way (decoder_blocks.py:171-172,
 * dropout2(activation(linear1(src))) with activation = relu), bf16, n % 8 == 0:
 *   y  = keep ? relu(h) * keep_scale : 0        keep: n mask bytes, NULL = keep all
 *   gh = y > 0 ? gy * keep_scale : 0            (y > 0  <=>  kept and h > 0)
 */
int transoar_relu_dropout_forward(const void* h, const unsigned char* keep, float keep_scale,
                                  const int* keep_seed, float keep_prob, void* y, long n, void* hip_stream);
int transoar_relu_dropout_backward(const void* gy, const void* y, float keep_scale, void* gh, long n,
                                   void* hip_stream);
int transoar_add_layernorm_partial_rows(void);

/*
 * Query of the first layer of the refine block (decoder_blocks.py:76-83 + with_pos_embed :157-158, applied to the
 * pyramid tokens themselves):  q16 = bf16(x16 + (pos_sine[s] + level_embed[l(s)]))  with x16 (rows, cols) bf16 --
 * the rounding points of the eager chain (fp32 sum of the positional terms, fp32 add, one rounding).
 * backward: the gradient of x16 is gq16 itself; partials (transoar_pos_query_partial_rows(), L, cols) fp32, written
 * completely, hold per workgroup the per-level column sums of gq16 (-> d level_embed: the caller sums over dim 0).
 */
int transoar_pos_query_forward(const void* x16, const float* pos_sine, const float* level_embed,
                               const int* level_start, int L, long S, void* q16, long rows, int cols,
                               void* hip_stream);
int transoar_pos_query_backward(const void* gq16, const int* level_start, int L, long S, float* partials,
                                long rows, int cols, void* hip_stream);
int transoar_pos_query_partial_rows(void);

/* Head of MSDeformAttn.forward (transoar/models/ops/modules/ms_deform_attn.py:114-128): from the stacked
 * projection proj (tokens, 4*M*L*P) bf16 = [sampling_offsets (M, L, P, 3) | attention_weights logits (M, L*P)],
 *     loc  (tokens, M, L, P, 3) fp32 = ref (ref_rows, L, 3)[token % ref_rows] + bf16(offset / bf16(W_l, H_l, D_l))
 *          (ref_rows = tokens, or the tokens of one batch element when the reference points are shared)
 *     attn (tokens, M, L, P)    fp32 = softmax over L*P of the logits
 * with the rounding points of the eager chain under bf16 autocast.  shapes (L, 3) int64 [D, H, W] on the device.
 * backward: g_proj (tokens, 4*M*L*P) bf16 from the gradients of loc and attn (ref receives none).  L*P <= 256. */
int transoar_sampling_head_forward(const void* proj, const float* ref, long ref_rows, const long* shapes, float* loc,
                                   float* attn, long tokens, int M, int L, int P, void* hip_stream);
int transoar_sampling_head_backward(const float* g_loc, const float* g_attn, const float* attn, const long* shapes,
                                    void* g_proj, long tokens, int M, int L, int P, void* hip_stream);

/*
 * LayerNorm over short rows (cols a multiple of 8, 8 <= cols <= 1536): the norms of the Swin encoder stages
 * (transoar/models/backbones/encoder_blocks.py:143-327, nn.LayerNorm over 48 .. 384 channels of 10^5 .. 10^6 tokens).
 *   forward:  y16 (rows, cols) bf16 = LayerNorm(x) * weight + bias, x bf16 or fp32; mean / rstd (rows) fp32 are kept for
 *             the backward
 *   backward: dx (x's type) from g16 (bf16); partials (transoar_ln_rows_partial_rows(), 2 * cols) fp32: every row holds a
 *             partial sum of [weight gradient | bias gradient] (column-sum them, e.g. transoar_rows_colsum_small).
 *             dx_add (x's type and shape) or NULL: a gradient that reaches x past the norm (the block's shortcut), added to dx
 *             in the same pass.
 */
int transoar_ln_rows_forward(const void* x, int x_is_bf16, const float* weight, const float* bias, float eps, void* y16,
                             float* mean, float* rstd, long rows, int cols, void* hip_stream);
int transoar_ln_rows_backward(const void* g16, const void* x, int x_is_bf16, const float* weight, const float* mean,
                              const float* rstd, const void* dx_add, void* dx, float* partials, long rows, int cols,
                              void* hip_stream);
int transoar_ln_rows_partial_rows(void);

int transoar_tokens_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TRANSOAR_TOKENS_H */
