"""ctypes binding and autograd shim of the fused masked cross-attention kernels (C ABI: include/transoar_attn.h;
kernels: csrc/attn.hip) -- SURVEY.md section 8, row f-1.

Reference semantics: FocusedAttn.forward, necks/focused_decoder.py:228-262 with the RoI mask of :138-159 / :243-247,
in the folded per-organ form of focused_decoder._roi_attention_folded: per (batch element, organ)

    ctx = softmax(mask(qf k^T)) v,      qf (R, C), k / v (L, C), C = 384, keys outside the organ's list masked.

Forward is one kernel (QK^T -> mask -> online softmax -> PV; plus a small combine pass when the keys are split across
workgroups); the backward recomputes P from the saved log-sum-exp in two kernels.  No fallback: the library must be built.
"""
import ctypes
import os

import torch

from . import _native  # noqa: F401  (torch's HIP runtime first)

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtransoar_attn.so")
ABI_VERSION = 2
CHANNELS = 384
MAX_ROWS = 512


def _load():
    if not os.path.exists(_LIB_PATH):
        raise _native.NativeLibraryError("%s is not built (python transoar_amd/_build.py)" % _LIB_PATH)
    lib = ctypes.CDLL(_LIB_PATH)
    i, p, l, z = ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_size_t
    lib.transoar_roi_attn_workspace_bytes.restype = z
    lib.transoar_roi_attn_workspace_bytes.argtypes = [i, i, i]
    lib.transoar_roi_attn_forward.restype = i
    lib.transoar_roi_attn_forward.argtypes = [p, p, p, p, p, p, p, p, z, i, i, i, l, i, i, p]
    lib.transoar_roi_attn_backward.restype = i
    lib.transoar_roi_attn_backward.argtypes = [p, p, p, p, p, p, p, p, p, p, p, z, i, i, i, l, i, i, p]
    lib.transoar_attn_abi_version.restype = i
    if lib.transoar_attn_abi_version() != ABI_VERSION:
        raise _native.NativeLibraryError("%s: ABI mismatch, rebuild" % _LIB_PATH)
    return lib


lib = _load()
ENABLED = os.environ.get("TRANSOAR_ROI_FUSED", "1") != "0"


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (what, rc))


def key_mask(pad):
    """pad (O, L) bool, True = padding -> (keybits (O, ceil(L/32)) int32: bit i of word t = key 32 t + i is masked, the
    bits past L set; n_tiles (O) int32: leading 32-key tiles that hold a real key).  Cached on the tensor (the RoI
    lists are fixed at construction), per (address, version): entries are only ever ADDED -- a captured graph may hold the
    address of an older entry's tensors, and a freed one is handed to the next allocation on its stream (DESIGN.md section
    12.4: the kernel then read its tile counts out of someone else's data)."""
    key = (pad.data_ptr(), pad._version, str(pad.device), tuple(pad.shape))
    cache = getattr(pad, "_transoar_key_mask", None)
    hit = None if cache is None else cache.get(key)
    if hit is not None:
        return hit
    n_org, n_keys = pad.shape
    tiles = (n_keys + 31) // 32
    full = torch.ones(n_org, tiles * 32, dtype=torch.bool, device=pad.device)
    full[:, :n_keys] = pad
    weights = (1 << torch.arange(32, dtype=torch.int64, device=pad.device))
    words = (full.view(n_org, tiles, 32).to(torch.int64) * weights).sum(-1)
    words = torch.where(words >= (1 << 31), words - (1 << 32), words).to(torch.int32).contiguous()
    live = ~full
    last = (live.to(torch.int64) * torch.arange(1, tiles * 32 + 1, device=pad.device)).amax(1)        # index of the last real key + 1
    n_tiles = ((last + 31) // 32).to(torch.int32).contiguous()
    try:
        if cache is None:
            cache = pad._transoar_key_mask = {}
        cache[key] = (words, n_tiles)          # a few hundred bytes per entry
    except Exception:        # noqa: BLE001  (a tensor subclass that refuses attributes: just do not cache)
        pass
    return words, n_tiles


def usable(qf, k_tok, v_tok):
    """Can the fused kernels take this attention?  bf16 on the GPU, dense (B, O, R / L, 384) operands, R <= 512."""
    return (ENABLED and qf.is_cuda and qf.dtype == torch.bfloat16 and k_tok.dtype == torch.bfloat16 and v_tok.dtype == torch.bfloat16
            and qf.dim() == 4 and k_tok.dim() == 4 and qf.shape[-1] == CHANNELS and k_tok.shape == v_tok.shape
            and k_tok.shape[-1] == CHANNELS and qf.shape[:2] == k_tok.shape[:2] and 0 < qf.shape[2] <= MAX_ROWS
            and (qf.shape[0] * qf.shape[1] * k_tok.shape[2] + 64) * 2 * CHANNELS < (1 << 31) - 1)


def pick_split(groups, rows):
    """Key splits per group so that the forward / dq grid (splits x row blocks of 128 x groups) is about one wave of
    workgroups on the chip's 256 CUs (one 512-register workgroup per CU)."""
    blocks = groups * ((rows + 127) // 128)
    return max(1, min(8, 256 // max(blocks, 1)))


def _stream():
    return torch.cuda.current_stream().cuda_stream


class RoiAttention(torch.autograd.Function):
    """ctx = softmax(mask(qf k^T)) v per (batch, organ).  The gradient of the keys is handed to `v_tok` together with the
    values' (callers pass keys = values + input-independent positions, focused_decoder._FoldedCore's convention); `k_tok`
    itself gets none."""

    @staticmethod
    def forward(ctx, qf, k_tok, v_tok, pad, n_split=None):
        b, o, r, c = qf.shape
        n_keys = k_tok.shape[2]
        qf, k_tok, v_tok = qf.contiguous(), k_tok.contiguous(), v_tok.contiguous()
        bits, n_tiles = key_mask(pad)
        g = b * o
        split = n_split or pick_split(g, r)
        out = torch.empty_like(qf)
        lse = torch.empty((g, r), dtype=torch.float32, device=qf.device)
        ws_bytes = lib.transoar_roi_attn_workspace_bytes(g, r, split)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=qf.device)
        with torch.cuda.device(qf.device):
            _check(lib.transoar_roi_attn_forward(qf.data_ptr(), k_tok.data_ptr(), v_tok.data_ptr(), bits.data_ptr(), n_tiles.data_ptr(),
                                                 out.data_ptr(), lse.data_ptr(), ws.data_ptr(), ws_bytes, g, o, r, n_keys, c, split,
                                                 _stream()), "transoar_roi_attn_forward")
        ctx.save_for_backward(qf, k_tok, v_tok, out, lse, bits, n_tiles)
        ctx.split = split
        return out

    @staticmethod
    def backward(ctx, dctx):
        qf, k_tok, v_tok, out, lse, bits, n_tiles = ctx.saved_tensors
        b, o, r, c = qf.shape
        n_keys = k_tok.shape[2]
        g = b * o
        dctx = dctx.to(torch.bfloat16).contiguous()
        dq = torch.empty_like(qf)
        dtok = torch.empty_like(v_tok)
        ws_bytes = lib.transoar_roi_attn_workspace_bytes(g, r, ctx.split)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=qf.device)
        with torch.cuda.device(qf.device):
            _check(lib.transoar_roi_attn_backward(qf.data_ptr(), k_tok.data_ptr(), v_tok.data_ptr(), out.data_ptr(), dctx.data_ptr(),
                                                  lse.data_ptr(), bits.data_ptr(), n_tiles.data_ptr(), dq.data_ptr(), dtok.data_ptr(),
                                                  ws.data_ptr(), ws_bytes, g, o, r, n_keys, c, ctx.split, _stream()),
                   "transoar_roi_attn_backward")
        return dq, None, dtok, None, None


def roi_attention(qf, k_tok, v_tok, pad, n_split=None):
    return RoiAttention.apply(qf, k_tok, v_tok, pad, n_split)
