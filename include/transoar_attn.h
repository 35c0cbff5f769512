/*
 * transoar_attn.h -- C ABI of the fused masked cross-attention of the Focused Decoder.
 *
 * Replaces, for one decoder layer, the score / mask / softmax / weighted-sum chain of FocusedAttn.forward
 * (transoar/models/necks/focused_decoder.py:228-262; mask from attn_area, :138-159 and :243-247) once the host has
 * gathered each organ's RoI tokens and folded the key / value projections into the queries
 * (transoar_amd/focused_decoder.py:_roi_attention_folded).  Per group g = (batch element, organ), g % O = organ:
 *
 *     S    = q[g] (R x C) . k[g]^T (C x L)          keys with their bit set in keybits[organ] are masked (-inf)
 *     ctx  = softmax_rows(S) . v[g] (L x C)         lse[g][row] = log sum exp of the row (natural log)
 *   backward:  P = exp(S - lse),  dP = dctx v^T,  dS = P o (dP - rowsum(dctx o ctx))
 *     dq   = dS k                                   (R x C)
 *     dtok = dS^T q + P^T dctx                      (L x C): the gradient of k and of v together (k = v + constant)
 *
 * All matrices bf16, row-major, dense: q / ctx / dctx / dq (G, R, C); k / v / dtok (G, L, C); C must be 384.
 * keybits (O, ceil(L / 32)) uint32, bit i of word t = key 32 t + i is padding -- the bits of keys >= L must be set;
 * n_tiles (O) int32 = number of leading 32-key tiles that hold at least one real key (the rest is skipped).
 * n_split: the keys of a group are divided among n_split workgroups in the forward and in the dq kernel (partial
 * results in `workspace`, transoar_roi_attn_workspace_bytes); the SAME value must be given to forward and backward
 * only as far as the workspace size goes -- results do not depend on it beyond fp32 rounding.
 * Device pointers, 16-byte aligned; asynchronous on `hip_stream`; returns 0, a hipError_t, or a negative code below.
 */
#ifndef TRANSOAR_ATTN_H
#define TRANSOAR_ATTN_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

enum {
  TRANSOAR_ATTN_OK = 0,
  TRANSOAR_ATTN_ERR_NULL = -1,
  TRANSOAR_ATTN_ERR_DIM = -2,          /* C != 384, G % O != 0, a size out of range */
  TRANSOAR_ATTN_ERR_WORKSPACE = -3
};

size_t transoar_roi_attn_workspace_bytes(int G, int R, int n_split);

int transoar_roi_attn_forward(const void* q, const void* k, const void* v, const unsigned* keybits, const int* n_tiles,
                              void* ctx, float* lse, void* workspace, size_t workspace_bytes, int G, int O, int R, long L,
                              int C, int n_split, void* hip_stream);

int transoar_roi_attn_backward(const void* q, const void* k, const void* v, const void* ctx, const void* dctx,
                               const float* lse, const unsigned* keybits, const int* n_tiles, void* dq, void* dtok,
                               void* workspace, size_t workspace_bytes, int G, int O, int R, long L, int C, int n_split,
                               void* hip_stream);

/*
 * Swin 3-D window attention (transoar/models/backbones/encoder_blocks.py:56-140):
 * per (window, head):  out = softmax(scale q k^T + bias[head] + mask[window % n_win]) v.
 *   qkv      (windows, n, 3, heads, 32) bf16: the output of the block's qkv projection, as it lies in memory
 *   bias     (heads, n, 128) fp32: relative-position bias, key axis padded to 128 (entries >= n are not read ... but the
 *            row must be addressable)
 *   maskbits (n_win, n, 4) uint32 or NULL: bit (key % 32) of word (key / 32) = the region labels of (row, key) differ,
 *            i.e. the reference's additive -100 (encoder_blocks.py:373-386); window w uses entry w % n_win
 *   out      (windows, n, heads * 32) bf16;  lse2 (windows, heads, n) fp32: log2-sum-exp2 of the rows (for the backward)
 * backward: dqkv (windows, n, 3, heads, 32) bf16 fully written; dbias (heads, n, 128) fp32 is ACCUMULATED into
 * (the caller zeroes it): the sum of dS over the windows.
 * n <= 128, head dimension 16 or 32 (`32` in the shapes above stands for it), scale > 0 (the row maximum is taken over
 * the unscaled scores); anything else returns TRANSOAR_ATTN_ERR_DIM.
 */
int transoar_win_attn_forward(const void* qkv, const float* bias, const unsigned* maskbits, void* out, float* lse2,
                              int windows, int n_win, int n, int heads, int head_dim, float scale, void* hip_stream);
int transoar_win_attn_backward(const void* qkv, const void* out, const void* dout, const float* lse2, const float* bias,
                               const unsigned* maskbits, void* dqkv, float* dbias, int windows, int n_win, int n, int heads,
                               int head_dim, float scale, void* hip_stream);

int transoar_attn_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
