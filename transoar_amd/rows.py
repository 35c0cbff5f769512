"""ctypes shim of the row gather / inverse-gather kernels (include/transoar_rows.h)."""
import ctypes
import os

import torch

from . import _native  # noqa: F401

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtransoar_rows.so")
if not os.path.exists(_LIB_PATH):
    raise _native.NativeLibraryError("%s is not built (python transoar_amd/_build.py)" % _LIB_PATH)
lib = ctypes.CDLL(_LIB_PATH)
_i, _p, _l = ctypes.c_int, ctypes.c_void_p, ctypes.c_long
lib.transoar_rows_gather.restype = _i
lib.transoar_rows_gather.argtypes = [_p, _p, _p, _i, _l, _l, _i, _p]
lib.transoar_rows_gather_axpy.restype = _i
lib.transoar_rows_gather_axpy.argtypes = [_p, _p, _p, _p, _p, _i, _l, _l, _i, _p]
lib.transoar_rows_pull_sum.restype = _i
lib.transoar_rows_pull_sum.argtypes = [_p, _p, _p, _p, _i, _l, _l, _i, _i, _p]
lib.transoar_rows_colsum.restype = _i
lib.transoar_rows_colsum.argtypes = [_p, _p, _p, _l, _i, _p]
lib.transoar_rows_colsum_workspace_floats.restype = _i
lib.transoar_rows_colsum_workspace_floats.argtypes = [_i]
lib.transoar_rows_colsum_small.restype = _i
lib.transoar_rows_colsum_small.argtypes = [_p, _p, _l, _i, _i, _p]


def colsum_usable(x):
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.bfloat16 and x.is_contiguous()
            and x.shape[1] % 8 == 0 and x.shape[1] <= 2048 and x.shape[0] >= 4096)


def colsum(x):
    """x (rows, cols) bf16 contiguous -> (cols,) fp32 column sums (fp32 accumulation)."""
    rows, cols = x.shape
    out = torch.empty(cols, dtype=torch.float32, device=x.device)
    work = torch.empty(lib.transoar_rows_colsum_workspace_floats(cols), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.transoar_rows_colsum(x.data_ptr(), out.data_ptr(), work.data_ptr(), rows, cols,
                                      torch.cuda.current_stream().cuda_stream)
    if rc:
        raise RuntimeError("transoar_rows_colsum failed with code %d" % rc)
    return out


def colsum_small_usable(x):
    return (x.is_cuda and x.dim() == 2 and x.is_contiguous() and x.data_ptr() % 16 == 0 and 0 < x.shape[0] <= 65536
            and ((x.dtype == torch.bfloat16 and x.shape[1] % 8 == 0) or (x.dtype == torch.float32 and x.shape[1] % 4 == 0)))


def colsum_small(x):
    """x (rows, cols) bf16 / fp32 contiguous, a short matrix -> (cols,) fp32 column sums in one launch."""
    rows, cols = x.shape
    out = torch.empty(cols, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.transoar_rows_colsum_small(x.data_ptr(), out.data_ptr(), rows, cols, int(x.dtype == torch.bfloat16),
                                            torch.cuda.current_stream().cuda_stream)
    if rc:
        raise RuntimeError("transoar_rows_colsum_small failed with code %d" % rc)
    return out


def colsum_any(x):
    """x.sum(0, dtype=float32) of a 2-D tensor on the kernel that fits its height (two-pass for >= 4096 bf16 rows, the
    one-launch kernel for short matrices and fp32 partials), torch's reduction otherwise."""
    if colsum_usable(x):
        return colsum(x)
    if colsum_small_usable(x):
        return colsum_small(x)
    return x.sum(0, dtype=torch.float32)


def usable(x):
    return (x.is_cuda and x.dim() == 3 and x.dtype in (torch.float32, torch.bfloat16)
            and (x.shape[2] * x.element_size()) % 16 == 0)


def row_dense(x):
    """(B, S, C) whose rows are dense and whose batch stride is a whole number of rows: a contiguous tensor, or one
    level cut out of the (N, S_total, C) pyramid -- what the gather kernel can read in place."""
    return (x.dim() == 3 and x.stride(2) == 1 and x.stride(1) == x.shape[2]
            and (x.shape[0] == 1 or (x.stride(0) % x.shape[2] == 0 and x.stride(0) >= x.shape[1] * x.shape[2]))
            and x.data_ptr() % 16 == 0)


def gather(x, index):
    """x (B,S,C) row-dense (see row_dense: the batch stride is passed as the kernel's S), index int32 (K,) -> (B,K,C)"""
    if not row_dense(x):
        x = x.contiguous()
    b, s, c = x.shape
    if b > 1:
        s = x.stride(0) // c            # rows between batch elements
    k = index.numel()
    out = torch.empty((b, k, c), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.transoar_rows_gather(x.data_ptr(), index.data_ptr(), out.data_ptr(), b, s, k,
                                      c * x.element_size(), torch.cuda.current_stream().cuda_stream)
    if rc:
        raise RuntimeError("transoar_rows_gather failed with code %d" % rc)
    return out


def gather_axpy(x, index, scale=None, resid=None):
    """bf16 x (B,S,C) row-dense, index int32 (K,), scale (B,) fp32 or None, resid (B,K,C) contiguous bf16 or None
    -> resid + scale[b] * x[:, index] (B,K,C) in one pass (negative index: a zero row)."""
    if x.dtype != torch.bfloat16 or (resid is not None and (resid.dtype != torch.bfloat16 or not resid.is_contiguous())):
        raise RuntimeError("gather_axpy: bf16 rows, contiguous residual")
    if not row_dense(x):
        x = x.contiguous()
    b, s, c = x.shape
    if b > 1:
        s = x.stride(0) // c
    k = index.numel()
    if resid is not None and tuple(resid.shape) != (b, k, c):
        raise RuntimeError("gather_axpy: residual must be (B, K, C)")
    out = torch.empty((b, k, c), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.transoar_rows_gather_axpy(x.data_ptr(), index.data_ptr(), None if scale is None else scale.data_ptr(),
                                           None if resid is None else resid.data_ptr(), out.data_ptr(), b, s, k,
                                           c * x.element_size(), torch.cuda.current_stream().cuda_stream)
    if rc:
        raise RuntimeError("transoar_rows_gather_axpy failed with code %d" % rc)
    return out


def pull_sum(g, inv_ptr, inv_idx, n_tokens):
    """g (B,K,C) contiguous -> (B,S,C): sum of the rows listed per token"""
    b, k, c = g.shape
    out = torch.empty((b, n_tokens, c), dtype=g.dtype, device=g.device)
    with torch.cuda.device(g.device):
        rc = lib.transoar_rows_pull_sum(g.data_ptr(), inv_ptr.data_ptr(), inv_idx.data_ptr(), out.data_ptr(), b,
                                        n_tokens, k, c * g.element_size(), 1 if g.dtype == torch.bfloat16 else 0,
                                        torch.cuda.current_stream().cuda_stream)
    if rc:
        raise RuntimeError("transoar_rows_pull_sum failed with code %d" % rc)
    return out
