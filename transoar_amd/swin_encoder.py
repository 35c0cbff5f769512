"""Swin 3-D window-attention encoder stages (BASELINE config #4:
``backbone.use_encoder_attn=True``).

Semantics of transoar/models/backbones/encoder_blocks.py:56-400 (Video-Swin blocks as wired by
attn_fpn.py:172-185): encoder stages 2..5 become ``depth`` pairs of window / shifted-window attention blocks
over 5x5x5 windows followed by a patch merge that halves the grid and doubles the channels; stages 0-1 stay
convolutional.  Parameter and buffer names follow the reference (checkpoint keys):

    _stages.#.blocks.#.{norm1,norm2}.{weight,bias}
    _stages.#.blocks.#.attn.{relative_position_bias_table, relative_position_index, qkv.*, proj.*}
    _stages.#.blocks.#.mlp.{fc1,fc2}.{weight,bias}
    _stages.#.downsample.{reduction.weight, norm.*}            (PatchMerging)
    _stages.#.downsample._reduction.{0.weight, 1.weight, 1.bias}   (ConvPatchMerging)

How it is computed is not the reference's: pad-to-window, cyclic shift and window partition are ONE token
gather through an index list built once per (grid, window, shift) -- and their inverses one gather back --
instead of pad + roll + view/permute/contiguous copies each way; the attention of all windows of all samples
runs as one fused scaled-dot-product call whose additive term (relative-position bias + shifted-window mask)
is assembled once per block.  Padding tokens are zero vectors that take part in the attention exactly as in
the reference (it pads AFTER norm1, encoder_blocks.py:158-164).
"""
import itertools
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import rows, win_attn
from . import tokens as fused_tokens
from .token_linear import gelu_mlp, gelu_mlp_usable, token_linear

MIN_TOKENS = int(os.environ.get("TRANSOAR_SWIN_MIN_TOKENS", "8192"))            # token matrices at least this tall take the hand-written GEMMs (token_linear)


def _fast(x):
    """The hand-written path of a Swin block: bf16 autocast on the GPU (fp32 / CPU: the plain torch formulation, which the
    golden parity tests pin)."""
    return x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16


def _norm16_fork(x, norm):
    """(x for the shortcut, LayerNorm(x) in bf16): one autograd node where the short-row kernel applies, so that the shortcut's
    gradient joins the norm's inside its backward kernel."""
    if fused_tokens.layernorm_rows_usable(x, norm) and x.requires_grad:
        return fused_tokens.layernorm_rows_fork(x, norm)
    return x, _norm16(x, norm)


def _norm16(x, norm):
    """LayerNorm rounded to bf16: the short-row kernel (tokens.layernorm_rows: 48 .. 512 channels) or torch + a cast."""
    if fused_tokens.layernorm_rows_usable(x, norm):
        return fused_tokens.layernorm_rows(x, norm)
    return norm(x).to(torch.bfloat16)


class _Rows(torch.autograd.Function):
    """x (B, S, C)[:, index] with the row-gather kernel of include/transoar_rows.h BOTH ways.  The window partition is a
    permutation of the tokens plus a few padding slots, so the adjoint of a gather is again a gather: through the inverse
    list, with -1 (= a zero row) where a source row is referenced by no slot.  (torch's indexing backward of the window
    partition took 21 ms per step in fp32; a pull over the CSR inverse, as the Focused Decoder's many-to-one key lists
    need, 15 ms on these 96-byte rows.)"""

    @staticmethod
    def forward(ctx, x, index, back_index):
        ctx.save_for_backward(back_index)
        return rows.gather(x, index)

    @staticmethod
    def backward(ctx, g):
        (back_index,) = ctx.saved_tensors
        return rows.gather(g.contiguous(), back_index), None, None


class _RowsAdd(torch.autograd.Function):
    """x + factor[b] * y[:, index]: the window merge (shift back + crop) of the attention branch, its stochastic-depth factor and
    the residual add in the gather's own pass; the adjoint for y is the same kernel through the inverse list."""

    @staticmethod
    def forward(ctx, x, y, index, back_index, factor):
        f = None if factor is None else factor.reshape(-1).float().contiguous()
        ctx.save_for_backward(back_index, f)
        return rows.gather_axpy(y, index, f, x.contiguous())

    @staticmethod
    def backward(ctx, g):
        back_index, f = ctx.saved_tensors
        g = g.contiguous()
        return g, rows.gather_axpy(g, back_index, f, None), None, None, None


class DropPath(nn.Module):
    """Stochastic depth per sample (timm.models.layers.DropPath as the reference uses it): in training a
    sample's branch is dropped with probability p and the survivors are scaled by 1/(1-p)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def factor(self, x):
        """The per-sample factor mask / keep, shaped (B, 1, ..., 1) like x -- or None when nothing is dropped."""
        if self.drop_prob == 0.0 or not self.training:
            return None
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return mask / keep

    def forward(self, x):
        f = self.factor(x)
        return x if f is None else x * f


def _add_path(x, y, drop_path):
    """x + drop_path(y) as one kernel (addcmul with the per-sample factor; its backward hands x's gradient through and scales
    y's once) instead of a multiply and an add over the token tensor each way."""
    f = drop_path.factor(y) if isinstance(drop_path, DropPath) else None
    return x + drop_path(y) if f is None else torch.addcmul(x, y, f)


class _BiasLookup(torch.autograd.Function):
    """table[index] with an index_add_ backward: the relative-position index is a fixed buffer, and torch's indexing backward
    (index_put_ with accumulate: sort + segmented sums) cost 80 us per block for 15 625 x heads values."""

    @staticmethod
    def forward(ctx, table, index):
        ctx.save_for_backward(index)
        ctx.rows = table.shape[0]
        return table[index]

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        out = torch.zeros((ctx.rows,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        if g.is_cuda and torch.are_deterministic_algorithms_enabled():
            # index_add_ accumulates with float atomics on the GPU: run-to-run differences in the last bits.  Under
            # torch.use_deterministic_algorithms the sorted accumulation of index_put_ (what table[index]'s own backward uses)
            return out.index_put_((index,), g.contiguous(), accumulate=True), None
        return out.index_add_(0, index, g.contiguous()), None


def effective_window(grid, window, shift=None):
    """Window (and shift) actually used on a grid: an axis not longer than the window is covered by one
    window and is not shifted (encoder_blocks.py:356-370)."""
    win = tuple(g if g <= w else w for g, w in zip(grid, window))
    if shift is None:
        return win
    return win, tuple(0 if g <= w else s for g, w, s in zip(grid, window, shift))


class _WindowLayout:
    """Index lists of one (grid, window, shift): which token sits at (window, position) after padding to a
    multiple of the window and rolling by -shift; the inverse; and the shifted-window mask."""

    def __init__(self, grid, win, shift, device):
        D, H, W = grid
        pad = tuple(-(-g // w) * w for g, w in zip(grid, win))
        Dp, Hp, Wp = pad
        n_tok = D * H * W
        # source token of every padded+rolled position: rolled[p] = padded[(p + shift) mod size]
        axes = [(torch.arange(n) + s) % n for n, s in zip(pad, shift)]
        d, h, w = torch.meshgrid(*axes, indexing="ij")
        src = (d * H + h) * W + w
        src = torch.where((d < D) & (h < H) & (w < W), src, torch.full_like(src, n_tok))     # n_tok = the zero row
        nd, nh, nw = Dp // win[0], Hp // win[1], Wp // win[2]
        src = src.view(nd, win[0], nh, win[1], nw, win[2]).permute(0, 2, 4, 1, 3, 5).reshape(-1)
        self.n_windows, self.n_per = nd * nh * nw, win[0] * win[1] * win[2]
        self.gather = src.to(device)
        # inverse: slot (window, position) that holds token t
        inv = torch.empty(n_tok + 1, dtype=torch.long)
        inv[src] = torch.arange(src.numel())
        self.scatter = inv[:n_tok].to(device)
        # the same two lists for the row kernel (int32; -1 = a zero row: the padding slots, so that no padding token has to be
        # appended to the token matrix) and the list each one's adjoint gathers through: the partition's adjoint is the merge
        # list, the merge's adjoint the partition list (padding slots are referenced by no token: -1 again)
        self.scatter32 = self.scatter.int()
        self.gather32 = torch.where(src < n_tok, src, torch.full_like(src, -1)).int().to(device)
        self.gather_back, self.scatter_back = self.scatter32, self.gather32
        self.mask = None
        self.mask_bits = None
        if any(s > 0 for s in shift):
            # region label of every padded position (3 slabs per axis), as seen after the roll
            # (encoder_blocks.py:373-386): tokens of one window attend only within equal labels
            label = torch.zeros(pad, dtype=torch.long)
            cuts = [((0, n - w_), (n - w_, n - s_), (n - s_, n)) for n, w_, s_ in zip(pad, win, shift)]
            for cnt, (sd, sh, sw) in enumerate(itertools.product(*cuts)):
                label[sd[0]:sd[1], sh[0]:sh[1], sw[0]:sw[1]] = cnt
            lab = label.view(nd, win[0], nh, win[1], nw, win[2]).permute(0, 2, 4, 1, 3, 5).reshape(self.n_windows, self.n_per)
            diff = lab[:, None, :] != lab[:, :, None]
            self.mask = torch.zeros(diff.shape).masked_fill_(diff, -100.0).to(device)
            if self.n_per <= win_attn.MAX_TOKENS:
                self.mask_bits = win_attn.mask_bits(self.mask)


_LAYOUTS = {}


def window_layout(grid, win, shift, device):
    key = (tuple(grid), tuple(win), tuple(shift), str(device))
    hit = _LAYOUTS.get(key)
    if hit is None:
        hit = _LAYOUTS[key] = _WindowLayout(tuple(grid), tuple(win), tuple(shift), device)
    return hit


class WindowAttention3D(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias, qk_scale, attn_drop, proj_drop):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, tuple(window_size), num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        wd, wh, ww = self.window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1), num_heads))
        # index of the pair (query position, key position) in the table: mixed radix of the offsets
        pos = torch.stack(torch.meshgrid(torch.arange(wd), torch.arange(wh), torch.arange(ww), indexing="ij")).flatten(1)
        rel = pos[:, :, None] - pos[:, None, :] + torch.tensor([wd - 1, wh - 1, ww - 1])[:, None, None]
        index = (rel[0] * (2 * wh - 1) + rel[1]) * (2 * ww - 1) + rel[2]
        self.register_buffer("relative_position_index", index)
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def position_bias(self, n):
        """(heads, n, n) relative-position bias of the first n positions of the window."""
        idx = self.relative_position_index[:n, :n].reshape(-1)
        table = self.relative_position_bias_table
        rows = _BiasLookup.apply(table, idx) if table.is_cuda else table[idx]
        return rows.view(n, n, -1).permute(2, 0, 1)

    def forward(self, x, mask=None, mask_bits=None):
        """x (B, nW, n, C) tokens per window; mask (nW, n, n) additive or None (mask_bits: the same as one bit per pair)."""
        b, nw, n, c = x.shape
        h = self.num_heads
        if (_fast(x) and x.dtype == torch.bfloat16 and c % h == 0 and c // h in win_attn.HEAD_DIMS and n <= win_attn.MAX_TOKENS
                and (mask is None or mask_bits is not None) and not (self.training and self.attn_drop.p > 0)
                and win_attn.ENABLED):
            # qkv projection -> ONE window-attention kernel over its output as it lies in memory -> output projection
            qkv = token_linear(x, self.qkv.weight, self.qkv.bias, force_hip=True, min_tokens=MIN_TOKENS)
            out = win_attn.window_attention(qkv.contiguous(), self.position_bias(n), mask_bits, h, self.scale)
            return self.proj_drop(token_linear(out, self.proj.weight, self.proj.bias, force_hip=True, min_tokens=MIN_TOKENS))
        q, k, v = self.qkv(x).view(b, nw, n, 3, h, c // h).permute(3, 0, 1, 4, 2, 5)      # each (B, nW, h, n, hd)
        bias = self.position_bias(n).to(q.dtype)[None]                                    # (1, h, n, n)
        if mask is not None:
            bias = bias + mask.to(q.dtype)[:, None]                                        # (nW, h, n, n)
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=bias,
                                             dropout_p=self.attn_drop.p if self.training else 0.0, scale=self.scale)
        return self.proj_drop(self.proj(out.transpose(2, 3).reshape(b, nw, n, c)))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        no_drop = not self.training or self.drop.p == 0.0
        if (_fast(x) and no_drop and isinstance(self.act, nn.GELU) and self.act.approximate == "none"
                and gelu_mlp_usable(x, self.fc1, self.fc2, MIN_TOKENS)):
            return gelu_mlp(x, self.fc1, self.fc2)          # the activation rides in the two GEMMs' epilogues
        if _fast(x):
            hid = self.act(token_linear(x, self.fc1.weight, self.fc1.bias, force_hip=True, min_tokens=MIN_TOKENS))
            return self.drop(token_linear(self.drop(hid), self.fc2.weight, self.fc2.bias, force_hip=True, min_tokens=MIN_TOKENS))
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class SwinBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio, qkv_bias, qk_scale, drop, attn_drop,
                 drop_path, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.window_size, self.shift_size = tuple(window_size), tuple(shift_size)
        assert all(0 <= s < w for s, w in zip(self.shift_size, self.window_size)), "shift_size must in 0-window_size"
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention3D(dim, self.window_size, num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x, grid):
        """x (B, D*H*W, C) tokens of a (D, H, W) grid."""
        b, n_tok, c = x.shape
        win, shift = effective_window(grid, self.window_size, self.shift_size)
        lay = window_layout(grid, win, shift, x.device)
        if _fast(x) and (c * 2) % 16 == 0:
            # bf16 tokens (what the qkv projection rounds them to anyway) through the row kernels both ways
            x, y = _norm16_fork(x, self.norm1)
            y = _Rows.apply(y, lay.gather32, lay.gather_back).view(b, lay.n_windows, lay.n_per, c)           # pad (zero rows) + shift + partition
            y = self.attn(y, lay.mask, lay.mask_bits)
            factor = self.drop_path.factor(x) if isinstance(self.drop_path, DropPath) else None
            if x.dtype == torch.bfloat16 and y.dtype == torch.bfloat16:
                # merge + shift back + crop, times the stochastic-depth factor, plus the shortcut: one pass
                x = _RowsAdd.apply(x, y.reshape(b, -1, c).contiguous(), lay.scatter32, lay.scatter_back, factor)
            else:
                y = _Rows.apply(y.reshape(b, -1, c).contiguous(), lay.scatter32, lay.scatter_back)            # merge + shift back + crop
                x = x + y if factor is None else torch.addcmul(x, y, factor)
            x, y = _norm16_fork(x, self.norm2)
            return _add_path(x, self.mlp(y), self.drop_path)
        y = self.norm1(x)
        y = torch.cat((y, y.new_zeros(b, 1, c)), dim=1)                      # row n_tok: the padding token
        y = y[:, lay.gather].view(b, lay.n_windows, lay.n_per, c)           # pad + shift + partition
        y = self.attn(y, lay.mask)
        y = y.reshape(b, -1, c)[:, lay.scatter]                              # merge + shift back + crop
        x = x + self.drop_path(y)
        return x + self.drop_path(self.mlp(self.norm2(x)))


_MERGE_LISTS = {}


def _merge_lists(grid, device):
    """Row lists of PatchMerging's 2x2x2 regrouping on a (D, H, W) token grid: slot (D', H', W', block) <- source token (or -1:
    the zero padding of an odd H / W), blocks in the reference's order (dd, dw, dh); and the inverse, slot of every source
    token (-1: the last plane of an odd D, which the reference's stride-2 slicing drops)."""
    key = (tuple(grid), str(device))
    hit = _MERGE_LISTS.get(key)
    if hit is None:
        D, H, W = grid
        d2, h2, w2 = D // 2, (H + 1) // 2, (W + 1) // 2
        od, oh, ow, dd, dw, dh = torch.meshgrid(torch.arange(d2), torch.arange(h2), torch.arange(w2), torch.arange(2), torch.arange(2),
                                                torch.arange(2), indexing="ij")
        sd, sh, sw = 2 * od + dd, 2 * oh + dh, 2 * ow + dw
        src = torch.where((sh < H) & (sw < W), (sd * H + sh) * W + sw, torch.full_like(sd, -1)).reshape(-1)
        back = torch.full((D * H * W,), -1, dtype=torch.long)
        ok = src >= 0
        back[src[ok]] = torch.arange(src.numel())[ok]
        hit = _MERGE_LISTS[key] = (src.int().to(device), back.int().to(device), (d2, h2, w2))
    return hit


class PatchMerging(nn.Module):
    """2x2x2 neighbours -> 8C channels -> LayerNorm -> Linear(8C, 2C, no bias) (encoder_blocks.py:298-327).
    Channel blocks in the reference's order: for d-offset 0,1: (h0,w0), (h1,w0), (h0,w1), (h1,w1).  Odd H/W are
    zero padded; D is not (sic)."""

    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(8 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(8 * dim)

    def forward(self, x):
        b, d, h, w, c = x.shape
        if _fast(x) and x.dtype == torch.bfloat16 and x.is_contiguous() and (c * 2) % 16 == 0 and d >= 2:
            # the 2x2x2 regrouping (pad, slice, permute, reshape: a strided copy each way) as the row gather of the window layout
            src, back, (d2, h2, w2) = _merge_lists((d, h, w), x.device)
            y = _Rows.apply(x.view(b, d * h * w, c), src, back).view(b, d2, h2, w2, 8 * c)
            return token_linear(_norm16(y, self.norm), self.reduction.weight, None, force_hip=True, min_tokens=MIN_TOKENS)
        if h % 2 or w % 2:
            x = F.pad(x, (0, 0, 0, w % 2, 0, h % 2))
            h, w = h + h % 2, w + w % 2
        x = x[:, : d - d % 2]                                   # stride-2 slicing of the reference drops a last odd plane
        x = x.view(b, d // 2, 2, h // 2, 2, w // 2, 2, c)       # (b, D, dd, H, dh, W, dw, c)
        x = x.permute(0, 1, 3, 5, 2, 6, 4, 7).reshape(b, d // 2, h // 2, w // 2, 8 * c)     # blocks ordered (dd, dw, dh)
        if _fast(x):
            return token_linear(_norm16(x.contiguous(), self.norm), self.reduction.weight, None, force_hip=True, min_tokens=MIN_TOKENS)
        return self.reduction(self.norm(x))


class ConvPatchMerging(nn.Module):
    def __init__(self, dim, bias=False, affine=True, eps=1e-05):
        super().__init__()
        self._reduction = nn.Sequential(
            nn.Conv3d(dim, dim * 2, kernel_size=2, stride=2, padding=0, bias=bias),
            nn.InstanceNorm3d(dim * 2, affine=affine, eps=eps), nn.ReLU(inplace=True))

    def forward(self, x):
        return self._reduction(x.permute(0, 4, 1, 2, 3)).permute(0, 2, 3, 4, 1)


class EncoderSwinBlock(nn.Module):
    """One Swin stage: ``depth`` blocks alternating plain / shifted windows, then the patch merge."""

    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio, qkv_bias, qk_scale, drop, attn_drop, drop_path,
                 downsample, norm_layer=nn.LayerNorm):
        super().__init__()
        self.window_size = tuple(window_size)
        self.shift_size = tuple(i // 2 for i in window_size)
        self.blocks = nn.ModuleList(
            SwinBlock(dim, num_heads, self.window_size, (0, 0, 0) if i % 2 == 0 else self.shift_size, mlp_ratio, qkv_bias,
                      qk_scale, drop, attn_drop, drop_path[i] if isinstance(drop_path, (list, tuple)) else drop_path,
                      norm_layer=norm_layer)
            for i in range(depth))
        self.downsample = None
        if downsample is not None:
            self.downsample = downsample(dim=dim) if downsample is ConvPatchMerging else downsample(dim=dim, norm_layer=norm_layer)

    def forward(self, x):
        """x (B, C, D, H, W) -> (B, 2C, D/2, H/2, W/2) (or (B, C, D, H, W) without a downsample)."""
        b, c, d, h, w = x.shape
        tokens = x.flatten(2).transpose(1, 2)                  # a free view for channels-last input
        for blk in self.blocks:
            tokens = blk(tokens, (d, h, w))
        y = tokens.view(b, d, h, w, c)
        if self.downsample is not None:
            y = self.downsample(y)
        return y.permute(0, 4, 1, 2, 3)
