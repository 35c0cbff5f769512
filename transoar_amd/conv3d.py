"""Autograd shim of the gfx950 3x3x3 convolution kernels (C ABI:
include/transoar_conv3d.h) and an ``nn.Conv3d`` subclass that uses them.

``Conv3dK3`` keeps nn.Conv3d's parameters (``weight`` (Cout,Cin,3,3,3), ``bias``)
so checkpoints of the reference's EncoderCnnBlock / FPN output convs
(encoder_blocks.py:28-48, attn_fpn.py:65-73) load unchanged.  On a GPU with
bf16 activations (the autocast training path) the convolution, its data
gradient and its weight gradient run on the hand-written implicit-GEMM
kernels, channels-last; otherwise (CPU tests, fp32/fp64 parity runs) it is
the stock PyTorch convolution.
"""
import ctypes
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _native  # noqa: F401  (torch's HIP runtime first)
from . import rows as _rows
from . import conv_gemm as _cg

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_PKG, "libtransoar_conv3d.so")
ABI_VERSION = 5


def _load():
    if not os.path.exists(_LIB_PATH):
        raise _native.NativeLibraryError("%s is not built (python transoar_amd/_build.py)" % _LIB_PATH)
    lib = ctypes.CDLL(_LIB_PATH)
    i, p = ctypes.c_int, ctypes.c_void_p
    lib.transoar_conv3d_k3_forward.restype = i
    lib.transoar_conv3d_k3_forward.argtypes = [p, p, p, p] + [i] * 8 + [p]
    lib.transoar_conv3d_k3_wgrad.restype = i
    lib.transoar_conv3d_k3_wgrad.argtypes = [p, p, p] + [i] * 10 + [p]
    lib.transoar_conv3d_k3_wgrad_lds.restype = i
    lib.transoar_conv3d_k3_wgrad_lds.argtypes = [p, p, p, i] + [i] * 10 + [p]
    lib.transoar_conv3d_k3_wgrad_tr.restype = i
    lib.transoar_conv3d_k3_wgrad_tr.argtypes = [p, p, p, i] + [i] * 10 + [p]
    lib.transoar_conv3d_c1_wgrad.restype = i
    lib.transoar_conv3d_c1_wgrad.argtypes = [p, p, p, i] + [i] * 5 + [p]
    lib.transoar_conv3d_c1_wgrad_tr.restype = i
    lib.transoar_conv3d_c1_wgrad_tr.argtypes = [p, p, p, i] + [i] * 5 + [p]
    lib.transoar_conv3d_k3_stat_rows.restype = i
    lib.transoar_conv3d_k3_stat_rows.argtypes = [i] * 7
    lib.transoar_conv3d_k3_forward_stats.restype = i
    lib.transoar_conv3d_k3_forward_stats.argtypes = [p, p, p, p, p] + [i] * 6 + [p]
    lib.transoar_conv3d_k3_forward_c1.restype = i
    lib.transoar_conv3d_k3_forward_c1.argtypes = [p, p, p, p] + [i] * 5 + [p]
    lib.transoar_conv3d_c1_forward.restype = i
    lib.transoar_conv3d_c1_forward.argtypes = [p, p, p] + [i] * 5 + [p]
    lib.transoar_layout_bf16.restype = i
    lib.transoar_layout_bf16.argtypes = [p, p, i, ctypes.c_long, i, i, p]
    lib.transoar_conv3d_abi_version.restype = i
    if lib.transoar_conv3d_abi_version() != ABI_VERSION:
        raise _native.NativeLibraryError("%s: ABI mismatch, rebuild" % _LIB_PATH)
    return lib


lib = _load()
CL3D = torch.channels_last_3d


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (what, rc))


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _Relayout(torch.autograd.Function):
    """The layout kernels as an autograd node: a change of memory format is the identity on the
    logical tensor, so the gradient passes through unchanged.  (Calling the raw kernel from a tracked
    forward returns a tensor without history and silently cuts the graph there.)"""

    @staticmethod
    def forward(ctx, t, to_channels_first):
        return _relayout_raw(t, to_channels_first)

    @staticmethod
    def backward(ctx, g):
        return g, None


def _relayout(t, to_channels_first):
    if t.requires_grad and torch.is_grad_enabled():
        return _Relayout.apply(t, to_channels_first)
    return _relayout_raw(t, to_channels_first)


def _relayout_raw(t, to_channels_first):
    n, c = t.shape[:2]
    v = t.shape[2] * t.shape[3] * t.shape[4]
    out = torch.empty(t.shape, dtype=t.dtype, device=t.device,
                      memory_format=torch.contiguous_format if to_channels_first else CL3D)
    with torch.cuda.device(t.device):
        _check(lib.transoar_layout_bf16(t.data_ptr(), out.data_ptr(), n, v, c, 1 if to_channels_first else 0,
                                        _stream()), "transoar_layout_bf16")
    return out


def to_ndhwc(t):
    """(N,C,D,H,W) logical -> bf16, physically NDHWC (channels_last_3d)."""
    t = t.to(torch.bfloat16)
    if t.is_contiguous(memory_format=CL3D):
        return t
    if t.is_cuda and t.is_contiguous() and t.shape[1] % 8 == 0:
        return _relayout(t, False)
    return t.contiguous(memory_format=CL3D)


def to_ncdhw(t):
    """physically NCDHW copy (or the tensor itself when it already is)."""
    if t.is_contiguous():
        return t
    if t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous(memory_format=CL3D) and t.shape[1] % 8 == 0:
        return _relayout(t, True)
    return t.contiguous()


_as_ndhwc = to_ndhwc


def _pack_taps(w):
    """(Cout, Cin, 3,3,3) -> (27, Cout, Cin) bf16, tap-major."""
    co, ci = w.shape[:2]
    return w.permute(2, 3, 4, 0, 1).reshape(27, co, ci).to(torch.bfloat16).contiguous()


def conv3d_k3_forward(x, wk, bias, stride, dilated_input=False):
    """x (N,Cin,D,H,W) NDHWC bf16; wk (27,Cout,Cin) bf16 -> (N,Cout,Do,Ho,Wo) NDHWC bf16."""
    n, ci, d, h, w = x.shape
    co = wk.shape[1]
    if dilated_input:
        od, oh, ow = 2 * d, 2 * h, 2 * w
    else:
        od, oh, ow = (d - 1) // stride + 1, (h - 1) // stride + 1, (w - 1) // stride + 1
    y = torch.empty((n, co, od, oh, ow), dtype=torch.bfloat16, device=x.device, memory_format=CL3D)
    with torch.cuda.device(x.device):
        _check(lib.transoar_conv3d_k3_forward(x.data_ptr(), wk.data_ptr(), bias.data_ptr() if bias is not None else None,
                                              y.data_ptr(), n, d, h, w, ci, co, stride, 1 if dilated_input else 0,
                                              _stream()), "transoar_conv3d_k3_forward")
    return y


def conv3d_k3_forward_c1(x, wk, bias):
    """x (N, 1, D, H, W) bf16 contiguous, wk (27, Cout, 8) -> (N, Cout, D, H, W) bf16 channels-last, stride 1"""
    n, _, d, h, w = x.shape
    cout = wk.shape[1]
    y = torch.empty((n, d, h, w, cout), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib.transoar_conv3d_k3_forward_c1(x.data_ptr(), wk.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                 y.data_ptr(), n, d, h, w, cout, _stream()), "conv3d_k3_forward_c1")
    return y.permute(0, 4, 1, 2, 3)


def conv3d_k3_forward_stats(x, wk, bias):
    """The halo-tile forward with the InstanceNorm statistics of its output in the epilogue: x (N, Cin, D, H, W) NDHWC bf16
    (Cin = 1: (N, 1, D, H, W) contiguous, wk (27, Cout, 8)) -> (y NDHWC bf16, part (rows, 2, 32) fp32) or None when the
    layer is not one of the halo-tile kernel's."""
    n, ci, d, h, w = x.shape
    co = wk.shape[1]
    rows = lib.transoar_conv3d_k3_stat_rows(n, d, h, w, ci, co, 1)
    if rows == 0:
        return None
    if ci == 1:
        y = torch.empty((n, d, h, w, co), dtype=torch.bfloat16, device=x.device).permute(0, 4, 1, 2, 3)
    else:
        y = torch.empty((n, co, d, h, w), dtype=torch.bfloat16, device=x.device, memory_format=CL3D)
    part = torch.empty((rows, 2, 32), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib.transoar_conv3d_k3_forward_stats(x.data_ptr(), wk.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                    y.data_ptr(), part.data_ptr(), n, d, h, w, ci, co, _stream()),
               "transoar_conv3d_k3_forward_stats")
    return y, part


def wgrad_operands(x, gy, stride):
    """Channels-first operands of the weight-gradient GEMM (K = voxels):
    gyT (Cout,N,Do,Ho,Wo) and xT3 (3,Cin,N,D,H,Wo), the three W-shifted (and for
    stride 2 W-decimated) copies of x."""
    ow = gy.shape[4]
    gy_t = gy.permute(1, 0, 2, 3, 4).contiguous()
    x_t = F.pad(x.permute(1, 0, 2, 3, 4).contiguous(), (1, 1))             # (Cin,N,D,H,W+2), zero halo
    x_t3 = torch.stack([x_t[..., kw: kw + stride * (ow - 1) + 1: stride] for kw in range(3)])
    return x_t3, gy_t


def conv3d_k3_wgrad(x, gy, stride):
    """x (N,Cin,D,H,W), gy (N,Cout,Do,Ho,Wo) bf16 -> dW (Cout,Cin,3,3,3) fp32."""
    n, ci, d, h, w = x.shape
    co, od, oh, ow = gy.shape[1], gy.shape[2], gy.shape[3], gy.shape[4]
    x_t3, gy_t = wgrad_operands(x, gy, stride)
    cin_p = (ci + 7) // 8 * 8
    dw = torch.zeros((27, co, cin_p), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib.transoar_conv3d_k3_wgrad(gy_t.data_ptr(), x_t3.data_ptr(), dw.data_ptr(), n, d, h, ow, ci,
                                            od, oh, ow, co, stride, _stream()), "transoar_conv3d_k3_wgrad")
    return dw[:, :, :ci].view(3, 3, 3, co, ci).permute(3, 4, 0, 1, 2)


C1_WGRAD_PARTIALS = 3200
C1_WGRAD_TR = os.environ.get("TRANSOAR_C1_WGRAD_TR", "1") != "0"
C1_WGRAD_TR_PARTIALS = 1024         # persistent: 4 workgroups per CU


def _colsum(partial):
    """Sum of the workgroups' partial tiles (groups, ...) fp32 -> (prod of the rest,) in one launch of the short-matrix column-sum
    kernel (torch's reduction needs 16-29 us for these)."""
    flat = partial.view(partial.shape[0], -1)
    return _rows.colsum_small(flat) if _rows.colsum_small_usable(flat) else flat.sum(0)


def conv3d_c1_wgrad(x, gy):
    """x (N,1,D,H,W) bf16 contiguous, gy (N,Cout,D,H,W) bf16 NDHWC -> dW (Cout,1,3,3,3) fp32."""
    n, _, d, h, w = x.shape
    co = gy.shape[1]
    tr = C1_WGRAD_TR and co % 8 == 0 and w % 64 == 0 and w <= 256          # coalesced loads + transposing LDS reads
    n_part = min(C1_WGRAD_TR_PARTIALS, n * d * h) if tr else min(C1_WGRAD_PARTIALS, max(1, (n * d * h + 3) // 4))
    partial = torch.empty((n_part, 32, 32), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        fn = lib.transoar_conv3d_c1_wgrad_tr if tr else lib.transoar_conv3d_c1_wgrad
        _check(fn(x.data_ptr(), gy.data_ptr(), partial.data_ptr(), n_part, n, d, h, w, co, _stream()), "transoar_conv3d_c1_wgrad")
    return _colsum(partial).view(32, 32)[:co, :27].reshape(co, 1, 3, 3, 3)


LDS_WGRAD_GROUPS = 512
LDS_WGRAD_TR = os.environ.get("TRANSOAR_WGRAD_TR", "1") != "0"      # second version: transposing LDS reads, 4 waves (conv3d_wgrad_tr.hpp)
LDS_WGRAD_TR_GROUPS = 512                                            # 2 workgroups per CU (202 VGPRs), 2 waves per SIMD


def conv3d_k3_wgrad_lds(x, gy):
    """x (N,Cin,D,H,W), gy (N,Cout,D,H,W) bf16 NDHWC, stride 1 -> dW (Cout,Cin,3,3,3) fp32
    (LDS-transposed MFMA kernel, one launch per block of <= 32 x 32 channels, W % 64 == 0)."""
    n, ci, d, h, w = x.shape
    co = gy.shape[1]
    dw = torch.empty((co, ci, 27), dtype=torch.float32, device=x.device)
    for co0 in range(0, co, 32):
        for ci0 in range(0, ci, 32):
            co_n, ci_n = min(32, co - co0), min(32, ci - ci0)
            groups = LDS_WGRAD_TR_GROUPS if LDS_WGRAD_TR else LDS_WGRAD_GROUPS
            fn = lib.transoar_conv3d_k3_wgrad_tr if LDS_WGRAD_TR else lib.transoar_conv3d_k3_wgrad_lds
            partial = torch.empty((groups, 27, 32, 32), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                _check(fn(x.data_ptr(), gy.data_ptr(), partial.data_ptr(), groups, n, d, h, w, ci, co, ci0, ci_n, co0, co_n, _stream()),
                       "transoar_conv3d_k3_wgrad_lds")
            dw[co0:co0 + co_n, ci0:ci0 + ci_n] = _colsum(partial).view(27, 32, 32)[:, :co_n, :ci_n].permute(1, 2, 0)
    return dw.view(co, ci, 3, 3, 3)


def lds_wgrad_supported(x, gy):
    ci, co = x.shape[1], gy.shape[1]
    return (ci % 8 == 0 and co % 8 == 0 and ci <= 64 and co <= 64 and x.shape[-1] % 64 == 0
            and x.is_contiguous(memory_format=CL3D) and gy.is_contiguous(memory_format=CL3D))


def c1_wgrad_supported(x, gy):
    return x.shape[1] == 1 and gy.shape[1] <= 32 and x.shape[-1] % 16 == 0 and x.is_contiguous()


class _Conv3dK3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, packs=None, want_stats=False):
        # one input channel: NDHWC == NCDHW; keep canonical strides so nothing downstream
        # (MIOpen's weight gradient) mistakes it for a channels-last problem
        xb = x.to(torch.bfloat16).contiguous() if x.shape[1] == 1 else _as_ndhwc(x)
        ci = xb.shape[1]
        wkt = None
        if want_stats:
            # the caller normalises the output next (EncoderCnnBlock): statistics in the conv epilogue where the halo-tile
            # kernel takes the layer; returns (y, part) -- part None when it does not
            res = None
            if stride == 1 and (ci == 1 or not _use_gemm(ci, weight.shape[0], stride)):
                wk = _pack_taps(F.pad(weight, (0, 0, 0, 0, 0, 0, 0, 7)) if ci == 1 else weight)
                res = conv3d_k3_forward_stats(xb, wk, bias.float() if bias is not None else None)
            if res is not None:
                ctx.save_for_backward(xb, weight, None)
                ctx.stride, ctx.has_bias, ctx.two = stride, bias is not None, True
                ctx.mark_non_differentiable(res[1])
                return res
        ctx.two = want_stats
        if ci == 1:
            # one input channel: through the MFMA implicit GEMM with the channel axis zero-padded to 8
            # (K = 27 taps x 8); the scalar stencil kernel (transoar_conv3d_c1_forward) is VALU-bound at a
            # seventh of this speed -- 1.6 ms vs 0.35 ms on the 2x160x160x256 volume
            n, _, d, h, w = xb.shape
            w8 = F.pad(weight, (0, 0, 0, 0, 0, 0, 0, 7))
            b32 = bias.float() if bias is not None else None
            if stride == 1 and weight.shape[0] <= 32 and d * h * w >= (1 << 16):
                # the LDS halo-tile kernel reads the one-channel volume itself (zero-extends while staging)
                y = conv3d_k3_forward_c1(xb, _pack_taps(w8), b32)
            else:
                x8 = torch.zeros((n, d, h, w, 8), dtype=torch.bfloat16, device=x.device)
                x8[..., 0] = xb.view(n, d, h, w)
                y = conv3d_k3_forward(x8.permute(0, 4, 1, 2, 3), _pack_taps(w8), b32, stride)
        elif _use_gemm(ci, weight.shape[0], stride):
            # 48 channels and up (and the strided 24 -> 48 layer): LDS-tiled implicit GEMM (csrc/conv_gemm.hip)
            # both filter packs in one pass over the fp32 weight when the backward will want the second one
            if packs is not None:
                wk, wkt = packs                 # packed for this weight version by conv_gemm.PackPlan (one launch per step)
            elif weight.requires_grad or x.requires_grad:
                wk, wkt = _cg.pack_both(weight)
            else:
                wk = _cg.pack_fwd(weight)
            y = _cg.conv_forward(xb, wk, bias.float() if bias is not None else None, stride)
        else:
            y = conv3d_k3_forward(xb, _pack_taps(weight), bias.float() if bias is not None else None, stride)
        ctx.save_for_backward(xb, weight, wkt)
        ctx.stride, ctx.has_bias = stride, bias is not None
        return (y, None) if want_stats else y

    @staticmethod
    def backward(ctx, gy, _gpart=None):
        # Every gradient runs on a hand-written kernel; there is no stock (MIOpen) branch.  A problem no kernel
        # covers raises: falling through to MIOpen's untuned 3-D bf16 kernels once cost a 1.7-second step.
        xb, weight, wkt = ctx.saved_tensors
        gyb = _as_ndhwc(gy)
        gx = gw = gb = None
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        ci, co = xb.shape[1], weight.shape[0]
        if need_x and _use_gemm(ci, co, ctx.stride):
            gx = _cg.conv_dgrad(gyb, wkt if wkt is not None else _cg.pack_dgrad(weight), ctx.stride, tuple(xb.shape[2:]))
        elif need_x:
            # data gradient = convolution of dy with the flipped, in/out-swapped filter (stride 2: over the dilated dy)
            wt = weight.flip(2, 3, 4).permute(2, 3, 4, 1, 0).reshape(27, weight.shape[1], weight.shape[0])
            gx = conv3d_k3_forward(gyb, wt.to(torch.bfloat16).contiguous(), None, 1, dilated_input=ctx.stride == 2)
        if need_w:
            rows = gyb.numel() // gyb.shape[1]
            if ctx.stride == 1 and c1_wgrad_supported(xb, gyb):
                gw = conv3d_c1_wgrad(xb, gyb)             # one input channel: MFMA over voxel chunks
            elif ctx.stride == 1 and lds_wgrad_supported(xb, gyb) and (max(ci, co) <= 32 or rows >= (1 << 21)):
                gw = conv3d_k3_wgrad_lds(xb, gyb)         # few channels, 10^7 voxels: LDS-transposed MFMA
            elif ci % 8 == 0 and co % 8 == 0:
                gw = _cg.conv_wgrad(xb, gyb, ctx.stride)  # voxel-major GEMM with transposing LDS reads (batch ranges above 2^21 rows)
            else:
                gw = conv3d_k3_wgrad(xb, gyb, ctx.stride)  # channels-first shifted copies: any channel count (raises on what it cannot address)
            gw = gw.to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            g2 = gyb.permute(0, 2, 3, 4, 1).reshape(-1, gyb.shape[1])        # channels-last: a view, rows = voxels
            gb = _rows.colsum_any(g2) if g2.is_contiguous() else gyb.float().sum(dim=(0, 2, 3, 4))
        return gx, gw, gb, None, None, None


def _use_gemm(cin, cout, stride):
    """Which forward / data-gradient kernel family: the LDS-tiled implicit GEMM from 48 channels up and for every strided
    layer with >= 8 input channels; the halo-tile / direct kernels of conv3d.hip for the full-resolution 24-channel layers."""
    return cin >= 8 and cin % 8 == 0 and cout % 8 == 0 and (stride == 2 or cin > 24 or cout > 32)


def hip_conv_supported(x, conv):
    return (x.is_cuda and x.dtype == torch.bfloat16 and conv.kernel_size == (3, 3, 3) and conv.padding == (1, 1, 1)
            and conv.stride in ((1, 1, 1), (2, 2, 2)) and conv.dilation == (1, 1, 1) and conv.groups == 1
            and (conv.in_channels % 8 == 0
                 or (conv.in_channels == 1 and conv.stride == (1, 1, 1) and conv.out_channels <= 64))
            and conv.out_channels % 8 == 0 and (x.shape[-1] // conv.stride[0]) % 8 == 0
            and (conv.stride == (1, 1, 1) or all(s % 2 == 0 for s in x.shape[2:])))


class Conv3dK3(nn.Conv3d):
    """nn.Conv3d(k=3, pad=1) whose bf16 GPU path is the hand-written kernel.
    ``enabled`` is a class switch so benchmarks can A/B against MIOpen;
    ``min_voxels``: below this many output voxels the grid is too small for the
    v1 kernel (no split-K) and the layer stays on MIOpen."""
    enabled = True
    min_voxels = 0          # every 3x3x3 layer of the model runs on the hand-written kernels (conv3d.hip, conv_gemm.hip)
    # channels-last maps stay channels-last into the stock (MIOpen) convolutions: its CK solvers are NDHWC natively,
    # and miopen_db/ holds find-db entries for the NDHWC keys of every layer of the flagship model (52.2 -> 51.4 ms
    # per step against converting to NCDHW first).  TRANSOAR_NDHWC_ALL=0: convert (the round-1 behaviour).
    ndhwc_everywhere = os.environ.get("TRANSOAR_NDHWC_ALL", "1") == "1"
    _packs, _packs_version = None, -1          # filter packs of the current weight version (conv_gemm.PackPlan), if any

    def uses_gemm(self):
        return self.kernel_size == (3, 3, 3) and _use_gemm(self.in_channels, self.out_channels, self.stride[0])

    def forward(self, x):
        amp = torch.is_autocast_enabled() and x.is_cuda and torch.get_autocast_gpu_dtype() == torch.bfloat16
        xb = x.to(torch.bfloat16) if (amp and x.dtype != torch.bfloat16) else x
        if Conv3dK3.enabled and hip_conv_supported(xb, self):
            s = self.stride[0]
            voxels = xb.shape[0] * (xb.shape[2] // s) * (xb.shape[3] // s) * (xb.shape[4] // s)
            if voxels >= Conv3dK3.min_voxels:
                packs = self._packs if self._packs_version == self.weight._version else None
                return _Conv3dK3.apply(xb, self.weight, self.bias, s, packs)
        # stock convolution (MIOpen on the GPU): NCDHW-contiguous input, see the note in backward
        if x.is_cuda and not Conv3dK3.ndhwc_everywhere:
            x = to_ncdhw(x)
        return super().forward(x)

    stats_in_epilogue = os.environ.get("TRANSOAR_CONV_STATS", "1") != "0"

    def forward_with_stats(self, x):
        """-> (y, part): forward(x) and, where the halo-tile kernel takes the layer (the full-resolution stride-1 layers of
        stage 0), the InstanceNorm statistics of y as per-workgroup partial sums (instnorm.instance_norm_relu's `part`);
        part is None everywhere else."""
        amp = torch.is_autocast_enabled() and x.is_cuda and torch.get_autocast_gpu_dtype() == torch.bfloat16
        xb = x.to(torch.bfloat16) if (amp and x.dtype != torch.bfloat16) else x
        if (Conv3dK3.enabled and Conv3dK3.stats_in_epilogue and self.stride == (1, 1, 1) and self.out_channels <= 32
                and hip_conv_supported(xb, self)):
            return _Conv3dK3.apply(xb, self.weight, self.bias, 1, None, True)
        return self.forward(x), None
