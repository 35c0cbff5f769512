/*
 * transoar_rows.h -- C ABI of the row gather / inverse gather used by the
 * Focused Decoder's per-organ key lists (host code: transoar_amd/focused_decoder.py;
 * reference semantics: the -inf mask of necks/focused_decoder.py:138-159,238-247).
 * Device pointers, 16-byte aligned rows, asynchronous on `hip_stream`; returns 0,
 * a hipError_t, -1 (NULL) or -2 (bad size).
 */
#ifndef TRANSOAR_ROWS_H
#define TRANSOAR_ROWS_H
#ifdef __cplusplus
extern "C" {
#endif

/* out[b][k][:] = src[b][index[k]][:]   src (B,S,row) out (B,K,row), index int32 (K).  S is only the distance between
 * batch elements in rows: a row-dense slice of a larger (B, S_total, row) matrix is read in place with S = S_total. */
int transoar_rows_gather(const void* src, const int* index, void* out, int B, long S, long K,
                         int row_bytes, void* hip_stream);

/* bf16 rows: out[b][k][:] = resid[b][k][:] + scale[b] * src[b][index[k]][:]  (fp32 arithmetic, one rounding).  The Swin block's
 * window merge + residual + stochastic-depth factor in one pass (reference: backbones/encoder_blocks.py, SwinTransformerBlock3D
 * forward: `x = shortcut + self.drop_path(x)` after window_reverse / roll / crop).  scale (B,) fp32 or NULL (= 1), resid (B,K,row)
 * dense or NULL (= 0); index < 0 reads a zero row. */
int transoar_rows_gather_axpy(const void* src, const int* index, const float* scale, const void* resid, void* out,
                              int B, long S, long K, int row_bytes, void* hip_stream);

/* out[b][s][:] = sum_{i in [inv_ptr[s], inv_ptr[s+1])} g[b][inv_idx[i]][:]   (fp32 or bf16 rows,
 * fp32 accumulation); inv_ptr int32 (S+1), inv_idx int32: the CSR inverse of `index`. */
int transoar_rows_pull_sum(const void* g, const int* inv_ptr, const int* inv_idx, void* out, int B,
                           long S, long K, int row_bytes, int is_bf16, void* hip_stream);

/* out[c] = sum_r x[r][c]: x (rows, cols) bf16, fp32 accumulation, out (cols) fp32; cols % 8 == 0, cols <= 2048.
 * workspace: transoar_rows_colsum_workspace_floats(cols) floats.  Replaces the bias-gradient reductions
 * grad.sum(0) of nn.Linear / nn.Conv3d(bias=True) under autocast. */
int transoar_rows_colsum(const void* x, float* out, float* workspace, long rows, int cols, void* hip_stream);
int transoar_rows_colsum_workspace_floats(int cols);

/* The same sum for SHORT matrices in one launch (a workgroup per 16-byte column group walks all rows): x (rows, cols)
 * bf16 (cols % 8 == 0) or fp32 (cols % 4 == 0), out (cols) fp32.  For the 10^3-row bias gradients of the Focused
 * Decoder's linears (necks/focused_decoder.py:12-59) and the per-wave partial sums of the token kernels. */
int transoar_rows_colsum_small(const void* x, float* out, long rows, int cols, int is_bf16, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
