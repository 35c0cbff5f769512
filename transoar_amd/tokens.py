"""Autograd shim of the fused residual-add + LayerNorm kernels
(include/transoar_tokens.h) for the token stream of the refine block.

``add_layernorm(x, r, norm, ...)`` returns what the reference's
``norm(x + r)`` returns under autocast (fp32) TOGETHER with its bf16 rounding
(what the next nn.Linear would cast it to) and, optionally, the bf16 query of the
next layer ``round(y + (pos_sine + level_embed[level]))`` -- one pass over the
tokens instead of add, layer_norm, two casts, the positional add and its cast.
"""
import ctypes
import os

import torch

from . import _native  # noqa: F401
from . import rows as _rows

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_PKG, "libtransoar_tokens.so")


def _load():
    if not os.path.exists(_LIB_PATH):
        raise _native.NativeLibraryError("%s is not built (python transoar_amd/_build.py)" % _LIB_PATH)
    lib = ctypes.CDLL(_LIB_PATH)
    i, p, lg, f = ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_float
    lib.transoar_add_layernorm_forward.restype = i
    lib.transoar_add_layernorm_forward.argtypes = [p, i, p, p, p, f, p, p, p, i, lg, p, p, p, p, lg, i, p, f, p, f, p]
    lib.transoar_add_layernorm_backward.restype = i
    lib.transoar_add_layernorm_backward.argtypes = [p, p, p, p, i, p, p, p, p, i, lg, p, p, p, lg, i, p, f, p, f, p]
    lib.transoar_relu_dropout_forward.restype = i
    lib.transoar_relu_dropout_forward.argtypes = [p, p, f, p, f, p, lg, p]
    lib.transoar_relu_dropout_backward.restype = i
    lib.transoar_relu_dropout_backward.argtypes = [p, p, f, p, lg, p]
    lib.transoar_add_layernorm_partial_rows.restype = i
    lib.transoar_sampling_head_forward.restype = i
    lib.transoar_sampling_head_forward.argtypes = [p, p, lg, p, p, p, lg, i, i, i, p]
    lib.transoar_sampling_head_backward.restype = i
    lib.transoar_sampling_head_backward.argtypes = [p, p, p, p, p, lg, i, i, i, p]
    lib.transoar_pos_query_forward.restype = i
    lib.transoar_pos_query_forward.argtypes = [p, p, p, p, i, lg, p, lg, i, p]
    lib.transoar_pos_query_backward.restype = i
    lib.transoar_pos_query_backward.argtypes = [p, p, i, lg, p, lg, i, p]
    lib.transoar_pos_query_partial_rows.restype = i
    lib.transoar_ln_rows_forward.restype = i
    lib.transoar_ln_rows_forward.argtypes = [p, i, p, p, ctypes.c_float, p, p, p, lg, i, p]
    lib.transoar_ln_rows_backward.restype = i
    lib.transoar_ln_rows_backward.argtypes = [p, p, i, p, p, p, p, p, p, lg, i, p]
    lib.transoar_ln_rows_partial_rows.restype = i
    lib.transoar_tokens_abi_version.restype = i
    if lib.transoar_tokens_abi_version() != 7:
        raise _native.NativeLibraryError("%s: ABI mismatch, rebuild" % _LIB_PATH)
    return lib


lib = _load()
PARTIAL_ROWS = lib.transoar_add_layernorm_partial_rows()
POS_QUERY_PARTIAL_ROWS = lib.transoar_pos_query_partial_rows()


def usable(x, r, cols):
    return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous()
            and (r is None or (r.dtype == torch.bfloat16 and r.is_contiguous() and r.shape == x.shape))
            and cols % 128 == 0 and cols // 128 in (1, 2, 3, 4, 6, 8))


def _ptr(t):
    return None if t is None else t.data_ptr()


def _keep_args(keep, scale):
    """(mask bytes, scale, seed, keep probability) of the C ABI from `keep`: None, a uint8 mask or an int32 seed."""
    if keep is None:
        return None, float(scale), None, 1.0
    if keep.dtype == torch.int32:
        return None, float(scale), keep.data_ptr(), 1.0 / float(scale)
    return keep.data_ptr(), float(scale), None, 1.0


class _AddLayerNorm(torch.autograd.Function):
    """(x, r, weight, bias, eps, pos_sine, level_embed, level_start, keep, keep_scale) -> (y32, y16, q16)"""

    @staticmethod
    def forward(ctx, x, r, weight, bias, eps, pos_sine, level_embed, level_start, keep, keep_scale):
        ctx.set_materialize_grads(False)          # an unused output (y32 of the last layer) costs no 360-MB zero gradient
        cols = x.shape[-1]
        rows = x.numel() // cols
        with_q = pos_sine is not None
        w32, b32 = weight.float().contiguous(), bias.float().contiguous()
        y32 = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        y16 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        q16 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if with_q else None
        stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        n_lvl = level_embed.shape[0] if with_q else 0
        s_tokens = pos_sine.shape[0] if with_q else rows
        le32 = level_embed.float().contiguous() if with_q else None
        with torch.cuda.device(x.device):
            rc = lib.transoar_add_layernorm_forward(
                x.data_ptr(), int(x.dtype == torch.bfloat16), _ptr(r), w32.data_ptr(), b32.data_ptr(), float(eps),
                _ptr(pos_sine), _ptr(le32), _ptr(level_start), n_lvl, s_tokens, y32.data_ptr(), y16.data_ptr(),
                _ptr(q16), stats.data_ptr(), rows, cols, *_keep_args(keep, keep_scale),
                torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("transoar_add_layernorm_forward failed with code %d" % rc)
        ctx.save_for_backward(x, r, w32, stats, level_start, keep)
        ctx.keep_scale = float(keep_scale)
        ctx.n_lvl, ctx.s_tokens, ctx.with_q = n_lvl, s_tokens, with_q
        ctx.param_dtype = weight.dtype
        ctx.le_dtype = level_embed.dtype if with_q else None
        return y32, y16, q16

    @staticmethod
    def backward(ctx, g32, g16, gq16):
        x, r, w32, stats, level_start, keep = ctx.saved_tensors
        cols = x.shape[-1]
        rows = x.numel() // cols
        g32 = None if g32 is None else g32.contiguous()
        g16 = None if g16 is None else g16.contiguous()
        gq16 = None if (gq16 is None or not ctx.with_q) else gq16.contiguous()
        x_bf16 = x.dtype == torch.bfloat16
        gx = torch.empty_like(x)
        gr = None
        if r is not None and ctx.needs_input_grad[1] and (not x_bf16 or keep is not None):
            gr = torch.empty_like(r)
        n_lvl = ctx.n_lvl if gq16 is not None else 0
        partials = torch.empty((PARTIAL_ROWS, 2 + n_lvl, cols), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.transoar_add_layernorm_backward(
                _ptr(g32), _ptr(g16), _ptr(gq16), x.data_ptr(), int(x_bf16), _ptr(r), w32.data_ptr(),
                stats.data_ptr(), _ptr(level_start), n_lvl, ctx.s_tokens, gx.data_ptr(), _ptr(gr),
                partials.data_ptr(), rows, cols, *_keep_args(keep, ctx.keep_scale),
                torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("transoar_add_layernorm_backward failed with code %d" % rc)
        sums = _rows.colsum_any(partials.view(PARTIAL_ROWS, -1)).view(2 + n_lvl, cols)     # one launch (rows.colsum_small)
        g_le = None
        if ctx.with_q and ctx.needs_input_grad[6]:
            g_le = (sums[2:] if n_lvl else torch.zeros(ctx.n_lvl, cols, device=x.device)).to(ctx.le_dtype)
        if r is not None and x_bf16 and keep is None:
            gr = gx
        return (gx, gr, sums[0].to(ctx.param_dtype), sums[1].to(ctx.param_dtype), None, None, g_le, None, None, None)


def dropout_mask(like, p):
    """Keep-mask bytes (1 = keep) for a dropout of probability p over a tensor shaped like `like`,
    from torch's generator (capture-safe), or None when nothing is dropped."""
    if p <= 0.0:
        return None
    return torch.empty(like.shape, dtype=torch.uint8, device=like.device).bernoulli_(1.0 - p)


SEEDED_DROPOUT = os.environ.get("TRANSOAR_DROPOUT_BYTES", "0") != "1"


def dropout_seed(like):
    """One int32 from torch's generator (capture-safe) on like's device: the kernels derive the keep-mask of the
    call from it by hashing the element index -- no mask tensor is written or read."""
    return torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int32, device=like.device)


def hashed_keep(seed, numel, keep_prob):
    """The mask the kernels derive from `seed` (int32 tensor of 1 element), as numel uint8 -- test reference of
    keep_pair() in csrc/tokens.hip."""
    m32 = 0xffffffff
    pair = torch.arange((numel + 1) // 2, dtype=torch.int64, device=seed.device)
    x = (pair * 0x9e3779b9 + (seed.to(torch.int64) & m32)) & m32
    x = x ^ (x >> 16)
    x = (x * 0x7feb352d) & m32
    x = x ^ (x >> 15)
    x = (x * 0x846ca68b) & m32
    x = x ^ (x >> 16)
    thr = min(max(int(keep_prob * 65536.0 + 0.5), 0), 65535)
    keep = torch.stack(((x & 0xffff) < thr, (x >> 16) < thr), dim=1).reshape(-1)[:numel]
    return keep.to(torch.uint8)


def add_layernorm(x, r, norm, pos_sine=None, level_embed=None, level_start=None, dropout=None):
    """-> (y32, y16, q16 or None).  x (..., C) fp32/bf16 residual stream, r bf16 branch or None,
    norm an nn.LayerNorm over C.  With pos_sine (S, C) fp32 (no grad), level_embed (L, C) and
    level_start (L,) int32, q16 = bf16(y + (pos_sine[s] + level_embed[level(s)])).
    dropout: the nn.Dropout that the reference applies to the branch first (applied inside the
    kernel when it is active: from a per-call seed, or a byte mask with TRANSOAR_DROPOUT_BYTES=1)."""
    keep, scale = None, 1.0
    if dropout is not None and dropout.training and dropout.p > 0.0 and r is not None:
        keep = dropout_seed(r) if SEEDED_DROPOUT else dropout_mask(r, dropout.p)
        scale = 1.0 / (1.0 - dropout.p)
    return _AddLayerNorm.apply(x, r, norm.weight, norm.bias, norm.eps, pos_sine, level_embed, level_start, keep, scale)


class _PosQuery(torch.autograd.Function):
    """(x16 (..., S, C) bf16, pos_sine (S, C) fp32, level_embed (L, C), level_start (L,) int32) -> q16 bf16"""

    @staticmethod
    def forward(ctx, x16, pos_sine, level_embed, level_start):
        cols = x16.shape[-1]
        rows = x16.numel() // cols
        le32 = level_embed.float().contiguous()
        q16 = torch.empty_like(x16)
        with torch.cuda.device(x16.device):
            rc = lib.transoar_pos_query_forward(x16.data_ptr(), pos_sine.data_ptr(), le32.data_ptr(),
                                                level_start.data_ptr(), le32.shape[0], pos_sine.shape[0],
                                                q16.data_ptr(), rows, cols, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("transoar_pos_query_forward failed with code %d" % rc)
        ctx.save_for_backward(level_start)
        ctx.dims = (rows, cols, le32.shape[0], pos_sine.shape[0], level_embed.dtype)
        return q16

    @staticmethod
    def backward(ctx, gq):
        level_start, = ctx.saved_tensors
        rows, cols, n_lvl, s_tokens, le_dtype = ctx.dims
        gq = gq.contiguous()
        g_le = None
        if ctx.needs_input_grad[2]:
            partials = torch.empty((POS_QUERY_PARTIAL_ROWS, n_lvl, cols), dtype=torch.float32, device=gq.device)
            with torch.cuda.device(gq.device):
                rc = lib.transoar_pos_query_backward(gq.data_ptr(), level_start.data_ptr(), n_lvl, s_tokens,
                                                     partials.data_ptr(), rows, cols,
                                                     torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                raise RuntimeError("transoar_pos_query_backward failed with code %d" % rc)
            g_le = _rows.colsum_any(partials.view(POS_QUERY_PARTIAL_ROWS, -1)).view(n_lvl, cols).to(le_dtype)
        return gq, None, g_le, None           # d q / d x = 1: the query's gradient IS the tokens' gradient


def pos_query(x16, pos_sine, level_embed, level_start):
    """bf16(x16 + (pos_sine[s] + level_embed[level(s)])) for contiguous bf16 tokens (..., S, C) in one pass; the
    backward hands the query's gradient on unchanged and reduces it per level for level_embed."""
    return _PosQuery.apply(x16, pos_sine, level_embed, level_start)


class _ReluDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, keep, scale):
        y = torch.empty_like(h)
        with torch.cuda.device(h.device):
            rc = lib.transoar_relu_dropout_forward(h.data_ptr(), *_keep_args(keep, scale), y.data_ptr(), h.numel(),
                                                   torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("transoar_relu_dropout_forward failed with code %d" % rc)
        ctx.save_for_backward(y)
        ctx.scale = float(scale)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, = ctx.saved_tensors
        gy = gy.contiguous()
        gh = torch.empty_like(y)
        with torch.cuda.device(y.device):
            rc = lib.transoar_relu_dropout_backward(gy.data_ptr(), y.data_ptr(), ctx.scale, gh.data_ptr(), y.numel(),
                                                    torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("transoar_relu_dropout_backward failed with code %d" % rc)
        return gh, None, None


def relu_dropout(h, dropout):
    """dropout(relu(h)) for a contiguous bf16 CUDA tensor (numel % 8 == 0) in one pass each way."""
    active = dropout.training and dropout.p > 0.0
    keep = (dropout_seed(h) if SEEDED_DROPOUT else dropout_mask(h, dropout.p)) if active else None
    return _ReluDropout.apply(h, keep, 1.0 / (1.0 - dropout.p) if active else 1.0)


class _SamplingHead(torch.autograd.Function):
    """(proj bf16 (N, Lq, 4*M*L*P), reference_points fp32 (N or 1, Lq, L, 3), shapes int64 (L, 3)) ->
    (locations fp32 (N, Lq, M, L, P, 3), attention weights fp32 (N, Lq, M, L, P))"""

    @staticmethod
    def forward(ctx, proj, reference_points, shapes, m, lv, pt):
        n, lq, _ = proj.shape
        loc = torch.empty((n, lq, m, lv, pt, 3), dtype=torch.float32, device=proj.device)
        attn = torch.empty((n, lq, m, lv, pt), dtype=torch.float32, device=proj.device)
        with torch.cuda.device(proj.device):
            rc = lib.transoar_sampling_head_forward(proj.data_ptr(), reference_points.data_ptr(),
                                                    reference_points.shape[0] * lq, shapes.data_ptr(),
                                                    loc.data_ptr(), attn.data_ptr(), n * lq, m, lv, pt,
                                                    torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("transoar_sampling_head_forward failed with code %d" % rc)
        ctx.save_for_backward(attn, shapes)
        ctx.dims = (m, lv, pt, proj.shape)
        return loc, attn

    @staticmethod
    def backward(ctx, g_loc, g_attn):
        attn, shapes = ctx.saved_tensors
        m, lv, pt, shape = ctx.dims
        g_loc, g_attn = g_loc.contiguous(), g_attn.contiguous()
        g_proj = torch.empty(shape, dtype=torch.bfloat16, device=attn.device)
        with torch.cuda.device(attn.device):
            rc = lib.transoar_sampling_head_backward(g_loc.data_ptr(), g_attn.data_ptr(), attn.data_ptr(), shapes.data_ptr(),
                                                     g_proj.data_ptr(), shape[0] * shape[1], m, lv, pt,
                                                     torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("transoar_sampling_head_backward failed with code %d" % rc)
        return g_proj, None, None, None, None, None


def sampling_head_raw(proj, reference_points, shapes, m, lv, pt):
    """The head's forward kernel without an autograd node: (locations, weights)."""
    n, lq, _ = proj.shape
    loc = torch.empty((n, lq, m, lv, pt, 3), dtype=torch.float32, device=proj.device)
    attn = torch.empty((n, lq, m, lv, pt), dtype=torch.float32, device=proj.device)
    with torch.cuda.device(proj.device):
        rc = lib.transoar_sampling_head_forward(proj.data_ptr(), reference_points.data_ptr(), reference_points.shape[0] * lq,
                                                shapes.data_ptr(), loc.data_ptr(), attn.data_ptr(), n * lq, m, lv, pt,
                                                torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("transoar_sampling_head_forward failed with code %d" % rc)
    return loc, attn


def sampling_head_backward_raw(g_loc, g_attn, attn, shapes, m, lv, pt, proj_shape):
    g_proj = torch.empty(proj_shape, dtype=torch.bfloat16, device=attn.device)
    with torch.cuda.device(attn.device):
        rc = lib.transoar_sampling_head_backward(g_loc.data_ptr(), g_attn.data_ptr(), attn.data_ptr(), shapes.data_ptr(),
                                                 g_proj.data_ptr(), proj_shape[0] * proj_shape[1], m, lv, pt,
                                                 torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("transoar_sampling_head_backward failed with code %d" % rc)
    return g_proj


def sampling_head_usable(proj, reference_points, shapes, m, lv, pt):
    return (proj.is_cuda and proj.dtype == torch.bfloat16 and proj.is_contiguous() and proj.dim() == 3
            and proj.shape[-1] == 4 * m * lv * pt and lv * pt <= 256
            and reference_points.dtype == torch.float32 and reference_points.is_contiguous()
            and not reference_points.requires_grad and tuple(reference_points.shape[1:]) == (proj.shape[1], lv, 3)
            and reference_points.shape[0] in (1, proj.shape[0])
            and shapes.dtype == torch.int64 and shapes.is_cuda and shapes.is_contiguous())


def sampling_head(proj, reference_points, shapes, m, lv, pt):
    """Sampling locations and attention weights of MSDeformAttn from the stacked projection, one kernel each way
    (instead of slice, divide, add, softmax and their backward chain with its concatenation)."""
    return _SamplingHead.apply(proj, reference_points, shapes, m, lv, pt)


def _ln_rows_forward(x, weight, bias, eps):
    cols = x.shape[-1]
    n_rows = x.numel() // cols
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(n_rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(n_rows, dtype=torch.float32, device=x.device)
    w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
    with torch.cuda.device(x.device):
        rc = lib.transoar_ln_rows_forward(x.data_ptr(), int(x.dtype == torch.bfloat16), w32.data_ptr(), b32.data_ptr(), float(eps),
                                          y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), n_rows, cols,
                                          torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("transoar_ln_rows_forward failed with code %d" % rc)
    return y, w32, mean, rstd


def _ln_rows_backward(g, x, w32, mean, rstd, g_add=None):
    cols = x.shape[-1]
    n_rows = x.numel() // cols
    g = g.to(torch.bfloat16).contiguous()
    if g_add is not None:
        g_add = g_add.to(x.dtype).contiguous()
    dx = torch.empty_like(x)
    partials = torch.empty((lib.transoar_ln_rows_partial_rows(), 2 * cols), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.transoar_ln_rows_backward(g.data_ptr(), x.data_ptr(), int(x.dtype == torch.bfloat16), w32.data_ptr(), mean.data_ptr(),
                                           rstd.data_ptr(), _ptr(g_add), dx.data_ptr(), partials.data_ptr(), n_rows, cols,
                                           torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("transoar_ln_rows_backward failed with code %d" % rc)
    from . import rows
    sums = rows.colsum_any(partials)
    return dx, sums[:cols], sums[cols:]


class _LayerNormRows(torch.autograd.Function):
    """nn.LayerNorm over the last axis of a (…, cols) tensor with 8 <= cols <= 512, cols % 8 == 0 (the Swin stages'
    norms, encoder_blocks.py:143-327), output bf16: one kernel each way (include/transoar_tokens.h: transoar_ln_rows_*)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = x.contiguous()
        y, w32, mean, rstd = _ln_rows_forward(x, weight, bias, eps)
        ctx.save_for_backward(x, w32, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, g):
        dx, gw, gb = _ln_rows_backward(g, *ctx.saved_tensors)
        return dx, gw, gb, None


class _LayerNormRowsFork(torch.autograd.Function):
    """(x, LayerNorm(x)) of a pre-norm residual block as ONE node: the gradient arriving at x past the norm (the shortcut) is
    added to the norm's input gradient inside the backward kernel instead of by autograd's accumulation pass over the tokens."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = x.contiguous()
        y, w32, mean, rstd = _ln_rows_forward(x, weight, bias, eps)
        ctx.save_for_backward(x, w32, mean, rstd)
        ctx.set_materialize_grads(False)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, g_x, g_y):
        x, w32, mean, rstd = ctx.saved_tensors
        if g_y is None:
            return g_x, None, None, None
        dx, gw, gb = _ln_rows_backward(g_y, x, w32, mean, rstd, g_x)
        return dx, gw, gb, None


def layernorm_rows_usable(x, norm):
    cols = x.shape[-1]
    return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and 8 <= cols <= 1536 and cols % 8 == 0
            and tuple(norm.normalized_shape) == (cols,) and norm.elementwise_affine and norm.bias is not None
            and x.data_ptr() % 16 == 0)


def layernorm_rows_fork(x, norm):
    """-> (x, norm(x) rounded to bf16); use the returned x for the block's shortcut (see _LayerNormRowsFork)."""
    return _LayerNormRowsFork.apply(x, norm.weight, norm.bias, norm.eps)


def layernorm_rows(x, norm):
    """norm(x) rounded to bf16 (what the following projection makes of it under bf16 autocast)."""
    return _LayerNormRows.apply(x, norm.weight, norm.bias, norm.eps)
