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

import torch
import torch.nn.functional as F
from torch import nn

from .ms_deform_attn import MSDeformAttn


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
        ffn = self.linear2(self.dropout2(self.activation(self.linear1(src))))
        return self.norm2(src + self.dropout3(ffn))


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
        tokens = torch.cat([f.flatten(2) for f in fmaps], dim=2).transpose(1, 2)          # (N, S, C)
        pos = torch.cat([p.flatten(2) + self.level_embed[lvl].view(1, -1, 1).to(p.dtype)
                         for lvl, p in enumerate(pos_embeds)], dim=2).transpose(1, 2)
        memory = self.refine_def_attn(tokens, spatial, starts, pos, ref)
        n, c = fmaps[0].shape[:2]
        return [m.transpose(1, 2).reshape(n, c, *shape)
                for m, shape in zip(memory.split(sizes, dim=1), shapes)]
