"""ctypes binding and autograd shim of the Swin 3-D window attention kernels (include/transoar_attn.h, csrc/attn.hip;
SURVEY.md section 8, row f-3).

Reference: WindowAttention3D.forward, transoar/models/backbones/encoder_blocks.py:56-140 -- per (window, head)
``softmax(scale q k^T + relative_position_bias + shifted-window mask) v`` -- as wired by SwinTransformerBlock3D
(:143-296).  One kernel each way over the qkv projection's output as it lies in memory; the (windows, heads, n, n)
score / mask tensors of the SDPA formulation never exist.  No fallback: the library must be built.
"""
import os

import torch
import torch.nn.functional as F

from . import roi_attn

lib = roi_attn.lib
HEAD_DIMS = (16, 32)          # the shipped configurations have 16 (48 / 96 / 192 / 384 channels, 3 / 6 / 12 / 24 heads)
MAX_TOKENS = 128
ENABLED = os.environ.get("TRANSOAR_WIN_ATTN", "1") != "0"

_i, _p, _f = roi_attn.ctypes.c_int, roi_attn.ctypes.c_void_p, roi_attn.ctypes.c_float
lib.transoar_win_attn_forward.restype = _i
lib.transoar_win_attn_forward.argtypes = [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p]
lib.transoar_win_attn_backward.restype = _i
lib.transoar_win_attn_backward.argtypes = [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p]


def mask_bits(mask):
    """Additive shifted-window mask (n_win, n, n) of 0 / -100 (encoder_blocks.py:373-386) -> (n_win, n, 4) int32:
    bit (key % 32) of word (key / 32) set where the mask is non-zero."""
    n_win, n, _ = mask.shape
    differ = torch.zeros(n_win, n, MAX_TOKENS, dtype=torch.bool, device=mask.device)
    differ[:, :, :n] = mask != 0
    weights = (1 << torch.arange(32, dtype=torch.int64, device=mask.device))
    words = (differ.view(n_win, n, 4, 32).to(torch.int64) * weights).sum(-1)
    return torch.where(words >= (1 << 31), words - (1 << 32), words).to(torch.int32).contiguous()


def usable(qkv, heads):
    """qkv (B, nW, n, 3 C) bf16 on the GPU, head dimension 16 or 32, n <= 128."""
    return (ENABLED and qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.dim() == 4 and qkv.shape[2] <= MAX_TOKENS
            and qkv.shape[3] % (3 * heads) == 0 and qkv.shape[3] // (3 * heads) in HEAD_DIMS and qkv.is_contiguous())


class _WindowAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, bias, bits, heads, scale):
        """qkv (B, nW, n, 3 C) bf16; bias (heads, n, n) fp32 (differentiable); bits (nW, n, 4) int32 or None."""
        b, n_win, n, c3 = qkv.shape
        windows = b * n_win
        bias_pad = F.pad(bias.detach().float(), (0, MAX_TOKENS - n)).contiguous()
        out = torch.empty((b, n_win, n, c3 // 3), dtype=torch.bfloat16, device=qkv.device)
        lse2 = torch.empty((windows, heads, n), dtype=torch.float32, device=qkv.device)
        with torch.cuda.device(qkv.device):
            roi_attn._check(lib.transoar_win_attn_forward(qkv.data_ptr(), bias_pad.data_ptr(), None if bits is None else bits.data_ptr(),
                                                          out.data_ptr(), lse2.data_ptr(), windows, n_win, n, heads, c3 // (3 * heads), float(scale),
                                                          roi_attn._stream()), "transoar_win_attn_forward")
        ctx.save_for_backward(qkv, out, lse2, bias_pad, bits)
        ctx.heads, ctx.scale = heads, float(scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse2, bias_pad, bits = ctx.saved_tensors
        b, n_win, n, c3 = qkv.shape
        windows = b * n_win
        dout = dout.to(torch.bfloat16).contiguous()
        dqkv = torch.empty_like(qkv)
        dbias = torch.zeros_like(bias_pad)
        with torch.cuda.device(qkv.device):
            roi_attn._check(lib.transoar_win_attn_backward(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse2.data_ptr(), bias_pad.data_ptr(),
                                                           None if bits is None else bits.data_ptr(), dqkv.data_ptr(), dbias.data_ptr(), windows,
                                                           n_win, n, ctx.heads, c3 // (3 * ctx.heads), ctx.scale, roi_attn._stream()),
                            "transoar_win_attn_backward")
        return dqkv, dbias[:, :, :n], None, None, None


def window_attention(qkv, bias, bits, heads, scale):
    """-> (B, nW, n, C) bf16: the heads' outputs side by side, the layout the output projection reads."""
    return _WindowAttention.apply(qkv, bias, bits, heads, scale)
