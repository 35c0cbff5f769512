"""Sine positional encoding over a 3-D feature map.

Same numbers as the reference's PositionEmbeddingSine3D
(transoar/models/position_encoding.py:10-51): per axis ``ceil(C/6)*2`` channels
of sin (first half) / cos (second half) at frequencies ``10000^(2*(i//2)/n)`` of the normalised
coordinate ``(idx+0.5)/(size+1e-6)*2*pi``; the three axis blocks are
concatenated in the order (second spatial axis, first spatial axis, third
spatial axis) and cut to C channels.  It depends only on the map's shape, so
it is built once per (shape, device, dtype) and cached -- the reference
rebuilds it with cumsums every forward.
"""
import math

import torch
from torch import nn


class PositionEmbeddingSine3D(nn.Module):
    def __init__(self, channels=64, temperature=10000, normalize=True, scale=None):
        super().__init__()
        if scale is not None and not normalize:
            raise ValueError("normalize should be True if scale is passed")
        self.orig_channels = channels
        self.channels = int(math.ceil(channels / 6) * 2)
        self.temperature = temperature
        self.normalize = normalize
        self.scale = 2 * math.pi if scale is None else scale
        self._cache = {}

    def _axis_table(self, size, device):
        """(size, self.channels) table for one axis."""
        idx = torch.arange(1, size + 1, dtype=torch.float32, device=device)   # cumsum of ones
        if self.normalize:
            idx = (idx - 0.5) / (size + 1e-6) * self.scale
        k = torch.arange(self.channels, dtype=torch.float32, device=device)
        freq = self.temperature ** (2 * torch.div(k, 2, rounding_mode="trunc") / self.channels)
        arg = idx[:, None] / freq
        # first half: sines of the even channels, second half: cosines of the odd
        # ones (the reference stacks on dim 4 of a 5-D tensor, i.e. BEFORE the
        # channel axis, so sin/cos are blocked, not interleaved)
        return torch.cat((arg[:, 0::2].sin(), arg[:, 1::2].cos()), dim=1)

    def _build(self, a, b, c, device):
        n = self.channels
        pos = torch.empty(3 * n, a, b, c, dtype=torch.float32, device=device)
        ta, tb, tc = (self._axis_table(s, device) for s in (a, b, c))
        pos[0:n] = tb.t()[:, None, :, None]          # second axis first (reference's "pos_y")
        pos[n:2 * n] = ta.t()[:, :, None, None]
        pos[2 * n:] = tc.t()[:, None, None, :]
        return pos[: self.orig_channels].contiguous()

    def forward(self, src):
        """src (N, C, A, B, Cdim) -> (N, orig_channels, A, B, Cdim) float32."""
        key = (tuple(src.shape[2:]), src.device)
        pos = self._cache.get(key)
        if pos is None:
            with torch.no_grad():
                pos = self._build(*src.shape[2:], device=src.device)
            self._cache[key] = pos
        out = pos[None].expand(src.shape[0], -1, -1, -1, -1)
        out._transoar_constant = True      # input independent: consumers may cache derived forms (a learned
        return out                         # encoding never carries this mark, whatever its requires_grad)


def is_constant(pos):
    """True for tensors produced by PositionEmbeddingSine3D (or cached forms derived from them)."""
    return bool(getattr(pos, "_transoar_constant", False))


class PositionEmbeddingLearned3D(nn.Module):
    """Learned absolute embedding (transoar/models/position_encoding.py:54-86):
    three 50-entry tables, one per axis."""

    def __init__(self, channels=128):
        super().__init__()
        self.orig_channels = channels
        n = int(math.ceil(channels / 6) * 2)
        self.row_embed = nn.Embedding(50, n)
        self.col_embed = nn.Embedding(50, n)
        self.depth_embed = nn.Embedding(50, n)
        for emb in (self.row_embed, self.col_embed, self.depth_embed):
            nn.init.uniform_(emb.weight)

    def forward(self, x):
        h, w, d = x.shape[-3:]
        dev = x.device
        col = self.col_embed(torch.arange(w, device=dev))     # indexed by the 2nd spatial axis
        row = self.row_embed(torch.arange(h, device=dev))     # 1st spatial axis
        dep = self.depth_embed(torch.arange(d, device=dev))   # 3rd spatial axis
        pos = torch.cat([col[None, :, None, :].expand(h, w, d, -1),
                         row[:, None, None, :].expand(h, w, d, -1),
                         dep[None, None, :, :].expand(h, w, d, -1)], dim=-1)
        pos = pos.permute(3, 0, 1, 2)[None].expand(x.shape[0], -1, -1, -1, -1)
        return pos[:, : self.orig_channels]
