/*
 * transoar_msda3d.h -- C ABI of the MI355X (gfx950) 3-D multi-scale deformable
 * attention operator.  This is the drop-in boundary: the two entry points
 * below are what the reference's pybind module "MultiScaleDeformableAttention"
 * exports and what its Python autograd wrapper binds:
 *
 *   reference export                         replaced by
 *   ---------------------------------------  ---------------------------------
 *   ms_deform_attn_forward                   transoar_msda3d_forward
 *     ops/src/vision.cpp:14                    (host launcher
 *     ops/src/ms_deform_attn.h:20-39            ms_deform_attn_cuda.cu:20-80)
 *   ms_deform_attn_backward                  transoar_msda3d_backward
 *     ops/src/vision.cpp:15                    (host launcher
 *     ops/src/ms_deform_attn.h:41-61            ms_deform_attn_cuda.cu:83-154)
 *
 * Differences from the reference boundary, on purpose:
 *   - plain pointers + sizes + a HIP stream; no ATen types.  The caller
 *     allocates every output (the reference allocates with at::zeros).
 *   - launch/argument errors are RETURNED (the reference only printf()s kernel
 *     launch errors, ms_deform_im2col_cuda.cuh:1119-1123).
 *   - one launch covers the whole batch; im2col_step (a chunking artefact of
 *     the reference launcher, .cu:50-75) has no meaning here and is not a
 *     parameter.  The Python shim keeps the argument for API compatibility.
 *   - 16-bit storage (bf16 / f16) is accepted with fp32 accumulation; the
 *     reference dispatches float/double only (.cu:64,135).
 *
 * Tensor contract (identical to the reference, .cu:40-48):
 *   value            (N, S, M, C)  contiguous, channel-last; S = sum_l D_l*H_l*W_l
 *   spatial_shapes   (L, 3) int64  [D, H, W] per level          DEVICE pointer
 *   level_start_idx  (L,)   int64  first row of each level      DEVICE pointer
 *   sampling_loc     (N, Lq, M, L, P, 3)  (x, y, z) = (W, H, D) axis, in [0,1]
 *   attn_weight      (N, Lq, M, L, P)
 *   out / grad_out   (N, Lq, M*C)
 * All device buffers must be 16-byte aligned (torch allocations are).
 * Everything is asynchronous on `stream`; no host synchronisation inside.  With TRANSOAR_MSDA3D_FORK
 * the backward forks part of its work onto an internal side stream and joins it back into `stream`
 * before it returns (graph-capture safe, but slower under graph replay; off by default).
 */
#ifndef TRANSOAR_MSDA3D_H
#define TRANSOAR_MSDA3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types of device buffers */
enum {
  TRANSOAR_F32 = 0,
  TRANSOAR_F64 = 1,
  TRANSOAR_BF16 = 2,
  TRANSOAR_F16 = 3
};

/* return codes: 0 = success; > 0 = a hipError_t from the launch;
 * < 0 = argument error detected on the host (see transoar_msda3d_strerror) */
enum {
  TRANSOAR_OK = 0,
  TRANSOAR_ERR_NULL = -1,        /* a required pointer is NULL            */
  TRANSOAR_ERR_DIM = -2,         /* a dimension is <= 0 or too large      */
  TRANSOAR_ERR_DTYPE = -3,       /* unsupported dtype combination         */
  TRANSOAR_ERR_ALIGN = -4,       /* a buffer is not 16-byte aligned       */
  TRANSOAR_ERR_LEVELS = -5,      /* L > TRANSOAR_MSDA3D_MAX_LEVELS        */
  TRANSOAR_ERR_WORKSPACE = -6,   /* workspace smaller than required       */
  TRANSOAR_ERR_CONST = -7,       /* device copy of the launch constants could not be made (first call for a shape inside a stream capture) */
  TRANSOAR_ERR_MODE = -8         /* TRANSOAR_MSDA3D_DETERMINISTIC asked for a form it does not cover (it covers the flagship form: 16-bit storage,
                                  * 64 channels, 4 points, <= 4 levels known on the host, queries = voxels of the pyramid) */
};

#define TRANSOAR_MSDA3D_MAX_LEVELS 8

/* flags (bit set) */
#define TRANSOAR_MSDA3D_FORCE_GENERIC 1u  /* skip the vectorised kernels   */
#define TRANSOAR_MSDA3D_NO_BRICK 4u        /* per-item / voxel-stationary kernels even where the LDS-tiled ones apply */
#define TRANSOAR_MSDA3D_FORK 8u             /* backward: run the coarse-level grad_value walk on an internal side stream */
#define TRANSOAR_MSDA3D_NO_MMA 16u            /* forward: LDS-tiled per-corner kernel instead of the matrix-core gather */
#define TRANSOAR_MSDA3D_MMA_Q32 32u           /* forward: round 2's matrix-core gather (32 queries per wave, fp32 weight block) instead of the point-column one */
#define TRANSOAR_MSDA3D_Q16 128u               /* forward: round 6's 16-queries-per-wave point-column gather (msda3d_q16.hpp) instead of round 3's (8 per wave).
                                                * Same results; slower as measured -- the probe of that carving, not the default. */
#define TRANSOAR_MSDA3D_DETERMINISTIC 64u      /* backward: bit-stable grad_value.  The sampling points are ordered by (cell, canonical
                                                * point index) with a stable radix sort instead of by the arrival order of atomic
                                                * cursors, and every level is accumulated by the brick-owner walk (no fp32 row atomics).
                                                * A test mode ("deterministic-mode backward"): ~4x the default backward's time;
                                                * the workspace query must be made with the same flags. */
#define TRANSOAR_MSDA3D_PULL_HEAD_MAJOR 2u /* schedule experiment: grad_value bricks walked head by head */

/*
 * Forward.  out[b,q,m*C+c] = sum_{l,p} attn[b,q,m,l,p] *
 *           trilinear(value_l[b,:,m,c], loc[b,q,m,l,p])   (zero padding,
 * align_corners=False; a point is skipped unless -1 < coord < size on all
 * axes).  `out` need not be initialised.
 *
 * value_dtype : dtype of value and out.
 * loc_dtype   : dtype of sampling_loc and attn_weight; must equal value_dtype,
 *               or be TRANSOAR_F32 when value_dtype is BF16/F16 (the layout
 *               bf16 autocast produces).
 * host_spatial_shapes: optional HOST copy of spatial_shapes (L*3 int64) or
 *               NULL.  The reference passes the shapes as a device tensor only;
 *               when the host also knows them the work is cut into 4x4x8-voxel
 *               bricks whose value rows / gradient rows live in LDS (16-bit
 *               storage, C = 64).  Must equal the device tensor.  With NULL the
 *               per-item kernels run: same results up to summation order.
 * Replaces ms_deform_attn_forward (ops/src/ms_deform_attn.h:20-39).
 */
int transoar_msda3d_forward(const void* value, const int64_t* spatial_shapes,
                            const int64_t* level_start_index,
                            const void* sampling_loc, const void* attn_weight,
                            void* out, int N, int S, int M, int C, int L,
                            int Lq, int P, int value_dtype, int loc_dtype,
                            const int64_t* host_spatial_shapes, unsigned flags,
                            void* hip_stream);

/*
 * Forward with the module's sampling head fused into the gather (the fast path of
 * MSDeformAttn.forward, ops/modules/ms_deform_attn.py:114-136: softmax over the
 * L*P logits of a head, sampling_locations = reference_points + offsets / (W, H, D),
 * then MSDeformAttnFunction).  Neither sampling_loc nor attn_weight is materialised:
 *   proj  (N*S, 4*M*L*P) bf16  the stacked projection of the query tokens,
 *         row = [sampling_offsets (M, L, P, 3) | attention logits (M, L*P)]
 *   ref   (ref_rows, L, 3) fp32 reference points (x, y, z); ref_rows = S (shared
 *         by the batch) or N*S
 * The arithmetic is that of transoar_sampling_head_forward followed by
 * transoar_msda3d_forward (the bf16 rounding points of the autocast chain).
 * Queries must be the voxels of the pyramid (Lq == S), 16-bit value storage,
 * C = 64, P = 4, L <= 4, host shapes required.  Returns TRANSOAR_ERR_DIM for
 * other forms (use the two-call path).
 */
int transoar_msda3d_forward_fused(const void* value, const void* proj,
                                  const float* ref, long ref_rows, void* out,
                                  int N, int S, int M, int C, int L, int P,
                                  int value_dtype,
                                  const int64_t* host_spatial_shapes,
                                  void* hip_stream);

/*
 * Backward.  Writes grad_value (shape and dtype of value), grad_sampling_loc
 * and grad_attn_weight (shapes of sampling_loc / attn_weight, loc_dtype)
 * completely; none of them needs to be initialised (the reference zero-fills
 * all three with at::zeros_like, .cu:122-124, and scatters with atomicAdd).
 *
 * `workspace` is a 16-byte aligned device scratch buffer of at least
 * transoar_msda3d_backward_workspace_bytes(...) bytes (same dims, dtypes and
 * flags); it holds the sampling points sorted by cell, from which grad_value
 * is accumulated brick by brick in LDS (fp32 row atomics into a scratch copy
 * only for pyramid levels with hundreds of points per voxel).  Its contents
 * are dead after the call's kernels have run on `stream`.
 * Replaces ms_deform_attn_backward (ops/src/ms_deform_attn.h:41-61).
 */
int transoar_msda3d_backward(const void* value, const int64_t* spatial_shapes,
                             const int64_t* level_start_index,
                             const void* sampling_loc, const void* attn_weight,
                             const void* grad_out, void* grad_value,
                             void* grad_sampling_loc, void* grad_attn_weight,
                             void* workspace, size_t workspace_bytes,
                             int N, int S, int M, int C, int L, int Lq, int P,
                             int value_dtype, int loc_dtype,
                             const int64_t* host_spatial_shapes, unsigned flags,
                             void* hip_stream);

/*
 * Backward with the sampling head's backward folded in (round 4).  Same inputs as transoar_msda3d_backward (the
 * locations / weights the head produced in the forward, ops/modules/ms_deform_attn.py:114-128); instead of
 * grad_sampling_loc / grad_attn_weight it writes grad_proj (N * Lq, 4 * M * L * P) bf16, the gradient of the stacked
 * [sampling_offsets | attention_weights] projection: offsets bf16(bf16(grad_loc) / bf16(W | H | D)), logits
 * a * (grad_attn - sum a * grad_attn) (softmax backward) -- the arithmetic of transoar_sampling_head_backward
 * (include/transoar_tokens.h), without the 2 x 360 MB of fp32 gradients in between.
 * Covers the matrix-core chain only: 16-bit value, fp32 locations, C = 64, P = 4, L = 4, queries = the pyramid's
 * voxels, host shapes given; anything else returns TRANSOAR_ERR_MODE.  Workspace as for transoar_msda3d_backward.
 */
int transoar_msda3d_backward_proj(const void* value, const int64_t* spatial_shapes,
                                  const int64_t* level_start_index, const void* sampling_loc,
                                  const void* attn_weight, const void* grad_out,
                                  void* grad_value, void* grad_proj, void* workspace,
                                  size_t workspace_bytes, int N, int S, int M, int C, int L,
                                  int Lq, int P, int value_dtype, int loc_dtype,
                                  const int64_t* host_spatial_shapes, unsigned flags,
                                  void* hip_stream);

/* Scratch bytes transoar_msda3d_backward needs (0 if the arguments are
 * invalid).  Pure host arithmetic. */
size_t transoar_msda3d_backward_workspace_bytes(int N, int S, int M, int C,
                                                int L, int Lq, int P,
                                                int value_dtype, int loc_dtype,
                                                unsigned flags);

/*
 * Optional per-kernel timing (used by bench.py for the roofline figures).
 * While enabled, every kernel the two entry points launch is bracketed by a
 * HIP event pair recorded on the caller's stream.  transoar_msda3d_profile_read
 * waits for the recorded pairs, adds their elapsed times per kernel kind into
 * total_ms[TRANSOAR_PROF_KINDS] / launches[TRANSOAR_PROF_KINDS] (both
 * overwritten) and forgets them.  Returns 0 or a hipError_t.
 */
enum {
  TRANSOAR_PROF_FWD = 0,          /* the forward gather: msda3d_fwd_pcm (flagship form) / _mma / _brick / _vec */
  TRANSOAR_PROF_BWD_QUERY = 1,    /* msda3d_bwd_query_mma (grad_loc, grad_attn or grad_proj, and the sort records) / _brick / _vec */
  TRANSOAR_PROF_CELL_COUNT = 2,   /* msda3d_cell_count_mma / msda3d_cell_count                  */
  TRANSOAR_PROF_SCAN = 3,         /* the three msda3d_scan_* launches   */
  TRANSOAR_PROF_CELL_FILL = 4,    /* fallback forms only: msda3d_cell_fill (the matrix-core chain writes its records from the query kernel) */
  TRANSOAR_PROF_PULL = 5,         /* msda3d_bwd_value_pull (voxel-stationary fallback) */
  TRANSOAR_PROF_FWD_GENERIC = 6,
  TRANSOAR_PROF_BWD_GENERIC = 7,
  TRANSOAR_PROF_VALUE_TILE = 8,   /* msda3d_bwd_value_tile_mma / _tile (+ msda3d_coarse_rows_store) */
  TRANSOAR_PROF_VALUE_CELLS = 9,  /* scratch memset + msda3d_bwd_value_cells_mma / _cells             */
  TRANSOAR_PROF_KINDS = 10
};
void transoar_msda3d_profile_enable(int on);
int transoar_msda3d_profile_read(double* total_ms, long* launches);

/* Human-readable text for a return code of the functions above. */
const char* transoar_msda3d_strerror(int code);

/* ABI version of this header: bumped on any signature change. */
int transoar_msda3d_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TRANSOAR_MSDA3D_H */
