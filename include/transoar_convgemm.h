/*
 * transoar_convgemm.h -- C ABI of the LDS-tiled implicit-GEMM 3x3x3 convolution kernels for MI355X (gfx950):
 * every 3^3 convolution of the backbone from 48 channels up, forward and both gradients
 * (reference layers: transoar/models/backbones/encoder_blocks.py:28-51 -- Conv3d(k 3, stride s, pad 1, no bias) of
 * the EncoderCnnBlock stages; transoar/models/backbones/attn_fpn.py:65-73,126 -- the FPN `out` Conv3d(k 3, pad 1)
 * with bias), which the reference runs through cuDNN via torch.nn.Conv3d.  Also the weight gradient of a token
 * projection (dW = dY^T X of nn.Linear: backbones/decoder_blocks.py:157-174, ops/modules/ms_deform_attn.py:109-140)
 * as the one-tap case of the same kernel.
 *
 * Layouts: activations NDHWC bf16 (torch channels_last_3d), i.e. rows = voxels, channels contiguous; filters
 * tap-major (27, N, K) bf16 with K (the contraction channels) contiguous.  All buffers 16-byte aligned, channel
 * counts multiples of 8.  Asynchronous on the given HIP stream; return 0 or an error code / hipError_t.
 *
 * Tap lists (taps_d / taps_h / taps_w), one per axis: bits [1:0] = number of entries (1..3); entry e:
 * bits [2+4e +: 2] = delta + 1 (source offset -1, 0, +1), bits [4+4e +: 2] = filter index t (0..2).
 * The taps of a launch are the Cartesian product of the three lists; filter slab = (t_d*3 + t_h)*3 + t_w.
 *   forward / weight gradient (stride s):   entries (-1,0) (0,1) (+1,2), source = s*m + delta
 *   data gradient of a stride-1 layer:      entries (+1,0) (0,1) (-1,2), source = m + delta, filter (27, Cin, Cout)
 *   data gradient of a stride-2 layer:      one launch per parity class of the dx voxels; per axis, parity 0:
 *                                           entry (0,1); parity 1: entries (+1,0) (0,2); source = m + delta in dy,
 *                                           output voxel = 2*m + parity
 */
#ifndef TRANSOAR_CONVGEMM_H
#define TRANSOAR_CONVGEMM_H

#ifdef __cplusplus
extern "C" {
#endif

enum {
  TRANSOAR_CONVGEMM_OK = 0,
  TRANSOAR_CONVGEMM_ERR_NULL = -1,
  TRANSOAR_CONVGEMM_ERR_DIM = -2
};

/*
 * y[out(m)][n] = sum_{tap} sum_{c < Cin} x[src(m, tap)][c] * wk[slab(tap)][n][c]  (+ bias[n])
 *   rows m run over (N, MD, MH, MW); src(m, tap) = src_stride*m + delta(tap) inside the source map (N, SD, SH, SW,
 *   Cin), zero outside; out(m) = out_stride*m + (opd, oph, opw) inside the output map (N, OD, OH, OW, Cout).
 *   split: the K steps are divided among `split` workgroups per tile.  With y32 != NULL every workgroup stores its
 *   fp32 partial tile to y32[split index][output row][n] (split maps of rows-of-the-OUTPUT-map x Cout floats, no
 *   initialisation needed, no atomics) and y is not written: follow with transoar_conv3d_finish (bias there).
 *   split > 1 requires y32.
 *   classes != 0: the data gradient of a stride-2 layer in ONE launch -- x is dy (N, SD, SH, SW, Cin = the layer's
 *   Cout), the output map (OD, OH, OW) is dx, wk is (27, layer Cin, layer Cout); MD.., strides, parity and the tap
 *   lists are derived per parity class inside (the arguments are ignored).
 */
int transoar_conv3d_igemm(const void* x, const void* wk, const float* bias, void* y, float* y32,
                          int N, int SD, int SH, int SW, int Cin, int Cout,
                          int MD, int MH, int MW, int src_stride,
                          int OD, int OH, int OW, int out_stride, int opd, int oph, int opw,
                          unsigned taps_d, unsigned taps_h, unsigned taps_w, int split, int classes, void* hip_stream);

/* y = bf16(sum of the `split` partial maps of y32 + bias) over rows x cout */
int transoar_conv3d_finish(const float* y32, const float* bias, void* y, long rows, int cout, int split, void* hip_stream);

/*
 * Data gradient of a stride-2 / pad-1 layer with few input channels (Cin <= 32, Cout = 16, 32 or 48: the 24 -> 48 layer
 * that opens stage 1, encoder_blocks.py:28-51 with stride 2): dx (N, D, H, W, Cin) from dy (N, OD, OH, OW, Cout) and
 * wkt (27, Cin, Cout); D = 2 OD or 2 OD - 1 (same for H, W).  One workgroup computes all eight parity classes of a
 * 4 x 8 x 16 dx tile from the dy halo in LDS and writes it as whole lines (HBM bound; transoar_conv3d_igemm with
 * classes = 1 is the general form).  Same result up to the order of the fp32 sums.
 */
int transoar_conv3d_dgrad_s2_halo(const void* dy, const void* wkt, void* dx, int N, int OD, int OH, int OW, int D, int H, int W,
                                  int Cin, int Cout, void* hip_stream);

/*
 * dw[co][ci][tap] (taps_out = 27: nn.Conv3d's weight layout (Cout, Cin, 3,3,3)) or dw[co][ci] (taps_out = 1: the one
 * tap of the lists, nn.Linear's (out, in)) = sum_m dy[m][co] * x[src(m, tap)][ci]          (fp32, written completely)
 *   rows m over (N, MD, MH, MW) = the voxels of dy (Cout channels); x is the source map (N, SD, SH, SW, Cin).
 *   chunks: number of voxel ranges the rows are split into (parallelism over the contraction axis); `part` is a
 *   scratch buffer of transoar_conv3d_wgrad_part_floats(Cin, Cout, chunks, taps_out) floats holding one partial map per
 *   chunk (no atomics); a second kernel sums them into dw.
 */
long transoar_conv3d_wgrad_part_floats(int Cin, int Cout, int chunks, int taps_out);
int transoar_conv3d_wgrad(const void* dy, const void* x, float* part, float* dw, int N, int SD, int SH, int SW, int Cin,
                          int Cout, int MD, int MH, int MW, int src_stride,
                          unsigned taps_d, unsigned taps_h, unsigned taps_w, int chunks, int taps_out, void* hip_stream);

/*
 * A token linear layer's weight AND bias gradient in one pass over dy (reference: the autograd of the nn.Linear layers of
 * backbones/encoder_blocks.py -- qkv, proj, fc1, fc2 of every Swin block): dw (Cout, Cin) = dy^T x as above (one tap), and
 * db (Cout) = sum_t dy[t][:] from the same MFMAs, as the product's column Cin against a column of ones that stands in the x
 * tile's padding.  dy (T, Cout), x (T, Cin) bf16; part as above (taps_out = 1); bias_part: chunks * Cout floats of scratch.
 * Cin must not be a multiple of the channel tile (64 when Cin, Cout <= 64, else 128): TRANSOAR_CONVGEMM_ERR_DIM otherwise.
 */
int transoar_linear_wgrad_bias(const void* dy, const void* x, float* part, float* dw, float* bias_part, float* db, int T, int Cin,
                               int Cout, int chunks, void* hip_stream);

/*
 * The same weight gradient (all 27 taps, taps_out = 27, forward tap lists) for layers with Cin, Cout <= 64, more than 32
 * channels on at least one side and MW % 64 == 0 (the 24 -> 48 / stride 2 and 48 -> 48 layers of stage 1): a workgroup
 * owns one filter plane kd and keeps the three x rows of its plane in an LDS ring, so that x and dy are fetched 3 times
 * instead of 27.  chunks = workgroups per plane (the grid is 3 x chunks: about one workgroup of 512 threads per CU);
 * part: transoar_conv3d_wgrad_part_floats(Cin, Cout, chunks, 27) floats.
 */
int transoar_conv3d_wgrad_ring(const void* dy, const void* x, float* part, float* dw, int N, int SD, int SH, int SW, int Cin,
                               int Cout, int MD, int MH, int MW, int src_stride, int chunks, void* hip_stream);

/* nn.Conv3d's fp32 weight (Cout, Cin, 3,3,3) -> wk (27, Cout, Cin) bf16 and, if wkt != NULL, wkt (27, Cin, Cout) bf16 */
int transoar_conv3d_pack(const float* w, void* wk, void* wkt, int Cout, int Cin, void* hip_stream);

/*
 * The same for many layers in ONE launch.  table (device memory): n_layers records of five 64-bit words
 * {w, wk, wkt (or 0), Cout | Cin << 32, first tile}, where a layer has ceil(Cout/32) * ceil(Cin/32) tiles numbered
 * consecutively from its first tile; total_tiles = the sum (= the grid).  All fp32 weights contiguous (Cout, Cin, 27).
 */
int transoar_conv3d_pack_many(const void* table, int n_layers, long total_tiles, void* hip_stream);

int transoar_convgemm_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
