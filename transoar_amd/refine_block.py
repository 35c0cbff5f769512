"""The FPN "refine" stage: multi-scale deformable self-attention over the
flattened feature pyramid -- the only caller of MSDeformAttn in the model.

Mirrors transoar/models/backbones/decoder_blocks.py (DecoderDefAttnBlock :12-97,
DefAttnTransformer :100-141, DefAttnLayer :143-177): same constructor
arguments, same parameter names (checkpoint keys ``refine_def_attn.layers.N.*``,
``level_embed``) and the same arithmetic.  Differences are host-side only:
the level geometry (spatial_shapes, level_start_index, reference points) is
built once per pyramid shape and cached instead of every forward, and the
tokens are kept in one (N, S, C) buffer written level by level.
"""
import copy
import os

import torch
import torch.nn.functional as F
from torch import nn

from .ms_deform_attn import MSDeformAttn
from .token_linear import fused_ffn, fused_ffn_usable, linear_relu_dropout, linear_relu_dropout_usable, token_linear

FUSED_FFN1 = os.environ.get("TRANSOAR_FUSED_FFN1", "1") != "0"      # linear1 + ReLU + dropout as one GEMM launch
from . import tokens as fused_tokens
from .position_encoding import is_constant


class _MapToTokens(torch.autograd.Function):
    """(N, C, D, H, W) bf16 NCDHW map -> (N, D*H*W, C) contiguous tokens through the
    layout kernel (and back in the backward).  The stock path
    ``f.flatten(2).transpose(1, 2)`` hands nn.Linear a transposed view, which
    costs hipBLASLt a 4x slower GEMM variant on the 117 000-token pyramid."""

    @staticmethod
    def forward(ctx, fmap):
        from .conv3d import to_ndhwc
        ctx.shape = fmap.shape
        n, c = fmap.shape[:2]
        return to_ndhwc(fmap).permute(0, 2, 3, 4, 1).reshape(n, -1, c)

    @staticmethod
    def backward(ctx, g):
        from .conv3d import Conv3dK3, to_ncdhw
        n, c, d, h, w = ctx.shape
        g = g.contiguous().view(n, d, h, w, c).permute(0, 4, 1, 2, 3)      # channels_last_3d view
        if Conv3dK3.ndhwc_everywhere and g.dtype == torch.bfloat16:
            # the producer is a channels-last convolution whose backward reads NDHWC: the view is its gradient as it
            # wants it (an NCDHW copy here was turned back by the convolution: two passes over every level)
            return g
        return to_ncdhw(g) if g.dtype == torch.bfloat16 else g.contiguous()


def map_to_tokens(fmap):
    if fmap.is_cuda and fmap.dtype == torch.bfloat16 and fmap.shape[1] % 8 == 0:
        return _MapToTokens.apply(fmap)
    return fmap.flatten(2).transpose(1, 2)


def tokens_to_map(tokens, shape):
    """(N, V, C) -> (N, C, D, H, W); a channels-last VIEW when the token rows are dense (no copy), which later
    flattens back to tokens for free."""
    n, _, c = tokens.shape
    if tokens.stride(2) == 1 and tokens.stride(1) == c:     # contiguous, or one level cut out of the (N, S, C) pyramid
        return tokens.view(n, *shape, c).permute(0, 4, 1, 2, 3)
    return tokens.transpose(1, 2).reshape(n, c, *shape)


def _activation(name):
    try:
        return {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}[name]
    except KeyError:
        raise RuntimeError("activation should be relu/gelu, not %s." % name)


class DefAttnLayer(nn.Module):
    """MSDeformAttn -> +res -> LN -> FFN -> +res -> LN (post-norm)."""

    def __init__(self, d_model, d_ffn, dropout, activation, n_levels, n_heads, n_points, use_cuda):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points, use_cuda)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _activation(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        query = src if pos is None else src + pos
        attn = self.self_attn(query, reference_points, src, spatial_shapes, level_start_index, padding_mask)
        src = self.norm1(src + self.dropout1(attn))
        hidden = self.dropout2(self.activation(token_linear(src, self.linear1.weight, self.linear1.bias)))
        ffn = token_linear(hidden, self.linear2.weight, self.linear2.bias)
        return self.norm2(src + self.dropout3(ffn))

    def forward_fused(self, x, x16, q16, reference_points, spatial_shapes, level_start_index, pos_pack=()):
        """Same layer on the fused token kernels (transoar_amd/tokens.py), bf16 autocast on the GPU.
        x: residual stream (fp32, or the bf16 backbone tokens for the first layer); x16 / q16: its
        bf16 rounding and the bf16 query round(x + pos).  pos_pack = (pos_sine, level_embed,
        level_start) asks for the next layer's query.  -> (y32, y16, q16 or None)"""
        attn = self.self_attn(q16, reference_points, x16, spatial_shapes, level_start_index)
        y32, y16, _ = fused_tokens.add_layernorm(x, attn.contiguous(), self.norm1, dropout=self.dropout1)
        if (self.activation is F.relu and fused_tokens.SEEDED_DROPOUT and FUSED_FFN1
                and fused_ffn_usable(y16, self.linear1.weight, self.linear2.weight)):
            # both layers as one autograd node: the gradient of ReLU + dropout rides in the epilogue of linear2's data-gradient
            # GEMM, the bias gradients in the weight-gradient passes (token_linear._FusedFFN)
            ffn = fused_ffn(y16, self.linear1, self.linear2, self.dropout2)
            return fused_tokens.add_layernorm(y32, ffn.contiguous(), self.norm2, *pos_pack, dropout=self.dropout3)
        if (self.activation is F.relu and fused_tokens.SEEDED_DROPOUT and FUSED_FFN1
                and linear_relu_dropout_usable(y16, self.linear1.weight)):
            # linear1 + bias + ReLU + dropout in ONE kernel (the K = 384 streaming GEMM's epilogue)
            hidden = linear_relu_dropout(y16, self.linear1.weight, self.linear1.bias, self.dropout2)
        else:
            hidden = token_linear(y16, self.linear1.weight, self.linear1.bias)
            if self.activation is F.relu and hidden.dtype == torch.bfloat16 and hidden.numel() % 8 == 0:
                hidden = fused_tokens.relu_dropout(hidden, self.dropout2)
            else:
                hidden = self.dropout2(self.activation(hidden))
        ffn = token_linear(hidden, self.linear2.weight, self.linear2.bias)
        return fused_tokens.add_layernorm(y32, ffn.contiguous(), self.norm2, *pos_pack, dropout=self.dropout3)


class DefAttnTransformer(nn.Module):
    def __init__(self, layer, num_layers):
        super().__init__()
        self.layers = nn.ModuleList(copy.deepcopy(layer) for _ in range(num_layers))
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes, device):
        """Voxel centres of every level as (x, y, z) in [0,1], shared by all
        levels: (1, S, L, 3).  (decoder_blocks.py:107-131 with valid_ratios==1)"""
        per_level = []
        for d, h, w in spatial_shapes.tolist():
            z = (torch.arange(d, dtype=torch.float32, device=device) + 0.5) / d
            y = (torch.arange(h, dtype=torch.float32, device=device) + 0.5) / h
            x = (torch.arange(w, dtype=torch.float32, device=device) + 0.5) / w
            grid = torch.stack((x[None, None, :].expand(d, h, w), y[None, :, None].expand(d, h, w),
                                z[:, None, None].expand(d, h, w)), dim=-1)
            per_level.append(grid.reshape(-1, 3))
        pts = torch.cat(per_level, 0)
        return pts[None, :, None, :].expand(1, -1, spatial_shapes.shape[0], -1).contiguous()

    def forward(self, src, spatial_shapes, level_start_index, pos=None, reference_points=None):
        if reference_points is None:
            reference_points = self.get_reference_points(spatial_shapes, src.device)
        out = src
        for layer in self.layers:
            out = layer(out, pos, reference_points, spatial_shapes, level_start_index)
        return out


class DecoderDefAttnBlock(nn.Module):
    def __init__(self, d_model, nhead, num_layers, dim_feedforward, dropout, feature_levels,
                 n_points, use_cuda=True, activation="relu"):
        super().__init__()
        self.d_model, self.nhead, self.feature_levels = d_model, nhead, feature_levels
        n_levels = len(feature_levels)
        layer = DefAttnLayer(d_model, dim_feedforward, dropout, activation, n_levels, nhead, n_points, use_cuda)
        self.refine_def_attn = DefAttnTransformer(layer, num_layers)
        self.level_embed = nn.Parameter(torch.empty(n_levels, d_model))
        self._geometry = {}
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        nn.init.normal_(self.level_embed)

    def _level_geometry(self, shapes, device):
        key = (shapes, device)
        geo = self._geometry.get(key)
        if geo is None:
            spatial = torch.as_tensor(shapes, dtype=torch.long, device=device)
            sizes = [d * h * w for d, h, w in shapes]
            starts = torch.as_tensor([sum(sizes[:i]) for i in range(len(sizes))], dtype=torch.long, device=device)
            ref = DefAttnTransformer.get_reference_points(spatial, device)
            geo = (spatial, starts, sizes, ref)
            self._geometry[key] = geo
        return geo

    def forward(self, fmaps, pos_embeds):
        """fmaps / pos_embeds: lists of (N, C, D_l, H_l, W_l) -> list of refined
        maps with the same shapes."""
        shapes = tuple(tuple(f.shape[2:]) for f in fmaps)
        spatial, starts, sizes, ref = self._level_geometry(shapes, fmaps[0].device)
        # token-major (N, S, C), contiguous: what the projections and the kernels read
        tokens = torch.cat([map_to_tokens(f) for f in fmaps], dim=1)
        if self._fused_ok(tokens, pos_embeds):
            memory = self._forward_fused(tokens, pos_embeds, shapes, spatial, starts, sizes, ref)
        else:
            pos = torch.cat([self._pos_tokens(p, lvl) + self.level_embed[lvl].view(1, 1, -1).to(p.dtype)
                             for lvl, p in enumerate(pos_embeds)], dim=1)
            memory = self.refine_def_attn(tokens, spatial, starts, pos, ref)
        # views into the token matrix (batch stride = the whole pyramid): no copies, the levels nobody reads cost nothing
        return [tokens_to_map(m, shape) for m, shape in zip(memory.split(sizes, dim=1), shapes)]

    bf16_out = os.environ.get("TRANSOAR_REFINE_FP32_OUT", "0") != "1"

    def _fused_ok(self, tokens, pos_embeds):
        return (tokens.is_cuda and tokens.dtype == torch.bfloat16 and tokens.is_contiguous()
                and torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16
                and all(is_constant(p) for p in pos_embeds)
                and fused_tokens.usable(tokens, None, self.d_model))

    def _forward_fused(self, tokens, pos_embeds, shapes, spatial, starts, sizes, ref):
        """The layer stack on the fused token kernels: the residual stream stays fp32 (what autocast's
        layer_norm returns), its bf16 rounding and the next layer's bf16 query come out of the same pass."""
        key = ("pos_sine", shapes, tokens.device)
        const = self._geometry.get(key)
        if const is None:
            # the sine encoding does not depend on the batch element or the input
            sine = torch.cat([self._pos_tokens(p, lvl)[0] for lvl, p in enumerate(pos_embeds)], dim=0)
            const = self._geometry[key] = (sine.float().contiguous(), starts.int().contiguous())
        pos_sine, starts32 = const
        pos_pack = (pos_sine, self.level_embed, starts32)
        layers = self.refine_def_attn.layers
        # first layer: query = round(tokens + (sine + level embedding)) in one pass (tokens.pos_query; the eager chain
        # wrote the 180-MB positional tensor and a 360-MB fp32 sum, and reduced their gradients level by level)
        q16 = fused_tokens.pos_query(tokens, pos_sine, self.level_embed, starts32)
        x, x16 = tokens, tokens
        for i, layer in enumerate(layers):
            last = i == len(layers) - 1
            x, x16, q16 = layer.forward_fused(x, x16, q16, ref, spatial, starts, () if last else pos_pack)
        # bf16_out: hand on the bf16 rounding of the last LayerNorm output (made in the same pass) instead of the fp32
        # tensor.  Under bf16 autocast every consumer of the refined maps (the Focused Decoder's token GEMMs, a
        # convolution) rounds them to bf16 first, so the values it sees are the same; what disappears is that cast,
        # its backward, and the 360-MB fp32 gradient of the whole pyramid that autograd zero-fills around the one
        # level the detector reads.  TRANSOAR_REFINE_FP32_OUT=1 returns the fp32 tensor (the reference's dtype).
        return x16 if DecoderDefAttnBlock.bf16_out else x

    def _pos_tokens(self, pos_map, lvl):
        """(N, C, D, H, W) positional map -> (N, V, C) tokens; the sine encoding
        is input independent, so its token form is cached per level and shape."""
        key = ("pos", lvl, tuple(pos_map.shape), pos_map.device)
        if not is_constant(pos_map):                    # learned encoding: no caching
            return pos_map.flatten(2).transpose(1, 2)
        hit = self._geometry.get(key)
        if hit is None:
            hit = self._geometry[key] = pos_map.flatten(2).transpose(1, 2).contiguous()
        return hit
