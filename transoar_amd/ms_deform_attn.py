"""Host-side mirror of the reference's operator surface for the hot path:

  MSDeformAttnFunction   autograd wrapper around the two MSDA entry points
                         (reference: ops/functions/ms_deform_attn_func.py:21-38)
  MSDeformAttn           the module with sub-modules named sampling_offsets,
                         attention_weights, value_proj, output_proj
                         (reference: ops/modules/ms_deform_attn.py:30-141;
                         the names fix checkpoint keys)

``use_cuda=True`` runs the gfx950 kernels through the C ABI.  ``use_cuda=False``
is the reference's "debug and test only" Python path
(ms_deform_attn_func.py:41-65); this package ships no CPU compute path, so that
flag only works after a test harness has injected a core with
``register_debug_core`` (tests and bench.py's cpu_baseline leg inject the
oracle's torch restatement).  Without an injected core it raises.
"""
import itertools
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import msda as MSDA
from .token_linear import token_linear
from .tokens import (sampling_head as _sampling_head, sampling_head_backward_raw as _head_bwd_raw,
                     sampling_head_raw as _head_raw, sampling_head_usable as _sampling_head_usable)

_debug_core = None


def _fused_usable(value, proj, reference_points, shapes, m, lv, pt, lq, s):
    """The fused head + gather entry covers: no gradient wanted, 16-bit value with 64 channels per head, 4 points, at
    most 4 levels, queries = the pyramid's voxels, host shapes available -- the refine block in evaluation."""
    if torch.is_grad_enabled() and (value.requires_grad or proj.requires_grad):
        return False
    return (MSDA.locality_hint and not (MSDA.flags & 0x35) and lq == s and pt == 4 and lv <= 4
            and value.dtype in (torch.bfloat16, torch.float16) and value.shape[-1] == 64 and value.is_contiguous()
            and _sampling_head_usable(proj, reference_points, shapes, m, lv, pt))


def register_debug_core(fn):
    """Install the callable used by ``MSDeformAttn(use_cuda=False)``:
    ``fn(value, spatial_shapes, sampling_locations, attention_weights) -> (N,Lq,M*C)``.
    Test/bench infrastructure only; returns the previous core."""
    global _debug_core
    prev, _debug_core = _debug_core, fn
    return prev


class MSDeformAttnFunction(Function):
    """value (N,S,M,C), spatial_shapes (L,3) int64 [D,H,W], level_start_index
    (L,) int64, sampling_locations (N,Lq,M,L,P,3) xyz, attention_weights
    (N,Lq,M,L,P), im2col_step -> (N,Lq,M*C).  Gradients flow to value,
    sampling_locations and attention_weights; no double backward."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index,
                sampling_locations, attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        out = MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                          sampling_locations, attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, starts, loc, attn = ctx.saved_tensors
        g_value, g_loc, g_attn = MSDA.ms_deform_attn_backward(
            value, shapes, starts, loc, attn, grad_output.contiguous(), ctx.im2col_step)
        return g_value, None, None, g_loc, g_attn, None


class _HeadGather(Function):
    """Sampling head + gather as ONE autograd node (round 4): forward = the head kernel and the gather, as before; the
    backward asks the operator for the gradient of the stacked projection directly (transoar_msda3d_backward_proj: the
    head's backward runs in the tail of the query kernel, the fp32 grad_loc / grad_attn never exist).  Forms that entry
    does not cover take the two-kernel backward inside the same node."""

    @staticmethod
    def forward(ctx, value, proj, reference_points, shapes, level_start, m, lv, pt, im2col_step):
        loc, attn = _head_raw(proj, reference_points, shapes, m, lv, pt)
        out = MSDA.ms_deform_attn_forward(value, shapes, level_start, loc, attn, im2col_step)
        ctx.save_for_backward(value, shapes, level_start, loc, attn)
        ctx.head = (m, lv, pt, tuple(proj.shape), im2col_step)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, starts, loc, attn = ctx.saved_tensors
        m, lv, pt, proj_shape, step = ctx.head
        grad_output = grad_output.contiguous()
        res = MSDA.ms_deform_attn_backward_proj(value, shapes, starts, loc, attn, grad_output, step)
        if res is None:
            g_value, g_loc, g_attn = MSDA.ms_deform_attn_backward(value, shapes, starts, loc, attn, grad_output, step)
            g_proj = _head_bwd_raw(g_loc.contiguous(), g_attn.contiguous(), attn, shapes, m, lv, pt, proj_shape)
        else:
            g_value, g_proj = res
        return g_value, g_proj, None, None, None, None, None, None, None


HEAD_GATHER = __import__("os").environ.get("TRANSOAR_HEAD_GATHER", "1") != "0"          # class switch (A/B and tests): the two-node path below is the reference's own structure


def _axis_directions(n_heads):
    """Sampling directions per head, in the lexicographic order of {-1,0,1}^3
    (the order the reference gets from torch.cartesian_prod,
    ms_deform_attn.py:67-73): 6 heads -> the axis unit vectors, 26 -> all
    non-zero lattice directions."""
    lattice = [v for v in itertools.product((-1.0, 0.0, 1.0), repeat=3)]
    if n_heads == 6:
        keep = [v for v in lattice if sum(abs(c) for c in v) == 1]
    elif n_heads == 26:
        keep = [v for v in lattice if any(c != 0 for c in v)]
    else:
        raise ValueError("Only nheads of value 26 or 6 are supported.")
    return torch.tensor(keep, dtype=torch.float32)


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, use_cuda=True):
        super().__init__()
        if d_model % n_heads:
            raise ValueError("d_model must be divisible by n_heads, but got %d and %d" % (d_model, n_heads))
        head_dim = d_model // n_heads
        if head_dim & (head_dim - 1):
            warnings.warn("MSDeformAttn: a power-of-two head dimension keeps the gfx950 kernels on "
                          "their vectorised path (C*elt in {128,256,512} bytes).")
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.use_cuda = use_cuda

        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 3)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        # offsets start as k voxels (k=1..P) along the head's direction, zero
        # weight; attention starts uniform (ms_deform_attn.py:63-91)
        step = torch.arange(1, self.n_points + 1, dtype=torch.float32)
        bias = _axis_directions(self.n_heads)[:, None, None, :] * step[None, None, :, None]
        bias = bias.expand(self.n_heads, self.n_levels, self.n_points, 3)
        with torch.no_grad():
            self.sampling_offsets.weight.zero_()
            self.sampling_offsets.bias.copy_(bias.reshape(-1))
            self.attention_weights.weight.zero_()
            self.attention_weights.bias.zero_()
            nn.init.xavier_uniform_(self.value_proj.weight)
            self.value_proj.bias.zero_()
            nn.init.xavier_uniform_(self.output_proj.weight)
            self.output_proj.bias.zero_()

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None):
        """query (N,Lq,d); reference_points (N,Lq,L,3) xyz in [0,1];
        input_flatten (N,S,d); input_spatial_shapes (L,3) [D,H,W];
        input_level_start_index (L,); input_padding_mask (N,S) True=pad
        -> (N,Lq,d)"""
        n, lq, _ = query.shape
        s = input_flatten.shape[1]
        if reference_points.shape[-1] != 3:
            raise ValueError("Last dim of reference_points must be 3 (x, y, z), got %d"
                             % reference_points.shape[-1])
        m, lv, pt = self.n_heads, self.n_levels, self.n_points

        value = token_linear(input_flatten, self.value_proj.weight, self.value_proj.bias)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], 0.0)
        value = value.view(n, s, m, self.d_model // m)

        # both projections read `query`: one GEMM over the stacked weights (the
        # parameters stay separate, as in the reference's state dict)
        n_off = self.sampling_offsets.out_features
        proj = token_linear(query, self.sampling_offsets.weight,
                            torch.cat((self.sampling_offsets.bias, self.attention_weights.bias)),
                            weight2=self.attention_weights.weight)
        if self.use_cuda and _fused_usable(value, proj, reference_points, input_spatial_shapes, m, lv, pt, lq, s):
            # no gradient wanted (evaluation / inference): the sampling head runs in the gather's prologue and
            # neither locations nor weights are materialised (transoar_msda3d_forward_fused)
            sampled = MSDA.ms_deform_attn_forward_fused(value, input_spatial_shapes, proj, reference_points, strict=False)
            if sampled is not None:         # None: a form the fused kernel does not cover (e.g. level sizes > 1000)
                return token_linear(sampled, self.output_proj.weight, self.output_proj.bias)
        if (self.use_cuda and HEAD_GATHER and lv == 4 and pt == 4 and torch.is_grad_enabled()
                and value.dtype in (torch.bfloat16, torch.float16)
                and _sampling_head_usable(proj, reference_points, input_spatial_shapes, m, lv, pt)):
            sampled = _HeadGather.apply(value, proj, reference_points, input_spatial_shapes, input_level_start_index,
                                        m, lv, pt, self.im2col_step)
            return token_linear(sampled, self.output_proj.weight, self.output_proj.bias)
        if self.use_cuda and _sampling_head_usable(proj, reference_points, input_spatial_shapes, m, lv, pt):
            locations, weights = _sampling_head(proj, reference_points, input_spatial_shapes, m, lv, pt)
        else:
            offsets = proj[..., :n_off].unflatten(-1, (m, lv, pt, 3))
            weights = F.softmax(proj[..., n_off:].unflatten(-1, (m, lv * pt)), dim=-1)
            weights = weights.view(n, lq, m, lv, pt)
            # offsets are in voxels of their level: divide by (W,H,D)
            whd = input_spatial_shapes.flip(-1).to(offsets.dtype)
            locations = reference_points[:, :, None, :, None, :] + offsets / whd[None, None, None, :, None, :]

        if self.use_cuda:
            sampled = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                                 locations, weights, self.im2col_step)
        else:
            if _debug_core is None:
                raise RuntimeError(
                    "MSDeformAttn(use_cuda=False): this package has no CPU path. The reference's "
                    "debug core lives in oracle/torch_ref.py; a test harness may inject it with "
                    "transoar_amd.ms_deform_attn.register_debug_core().")
            sampled = _debug_core(value, input_spatial_shapes, locations, weights)
        return token_linear(sampled, self.output_proj.weight, self.output_proj.bias)
