"""ctypes binding of the gfx950 token GEMM (include/transoar_gemm.h): ``linear_nt(x, w, bias, relu)`` =
``relu?(x @ w.T + bias)`` for bf16/f16 operands with fp32 accumulation.  No fallback: raises if the library is
missing."""
import ctypes
import os

import torch

from . import _native  # noqa: F401  (torch's HIP runtime first)

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_PKG, "libtransoar_gemm.so")
ABI_VERSION = 4
STREAM_SQUARE = os.environ.get("TRANSOAR_GEMM_STREAM_SQUARE", "0") == "1"
STREAM_N384 = os.environ.get("TRANSOAR_GEMM_N384", "0") == "1"
STREAM = os.environ.get("TRANSOAR_GEMM_STREAM", "1") != "0"      # the K = 384 / N = 384 streaming kernels (csrc/gemm_stream.hip)
_DT = {torch.float32: 0, torch.bfloat16: 2, torch.float16: 3}


def _load():
    if not os.path.exists(_LIB_PATH):
        raise _native.NativeLibraryError("%s is not built (python transoar_amd/_build.py)" % _LIB_PATH)
    lib = ctypes.CDLL(_LIB_PATH)
    i, p = ctypes.c_int, ctypes.c_void_p
    lib.transoar_gemm_nt.restype = i
    lib.transoar_gemm_nt.argtypes = [p, p, p, p] + [i] * 9 + [p]
    lib.transoar_gemm_nt_gelu.restype = i
    lib.transoar_gemm_nt_gelu.argtypes = [p, p, p, p, p] + [i] * 7 + [p]
    lib.transoar_gemm_k384.restype = i
    lib.transoar_gemm_k384.argtypes = [p, p, p, p, i, i, i, p]
    lib.transoar_gemm_k384_drop.restype = i
    lib.transoar_gemm_k384_drop.argtypes = [p, p, p, p, i, i, i, p, ctypes.c_float, ctypes.c_float, p]
    lib.transoar_gemm_k384_gate.restype = i
    lib.transoar_gemm_k384_gate.argtypes = [p, p, p, p, i, i, ctypes.c_float, p]
    lib.transoar_gemm_n384.restype = i
    lib.transoar_gemm_n384.argtypes = [p, p, p, p, i, i, p]
    lib.transoar_gemm_wgrad384_chunks.restype = i
    lib.transoar_gemm_wgrad384_chunks.argtypes = [i, i]
    lib.transoar_gemm_wgrad384.restype = i
    lib.transoar_gemm_wgrad384.argtypes = [p, p, p, p, i, i, i, i, p]
    lib.transoar_gemm_wgrad384_bias.restype = i
    lib.transoar_gemm_wgrad384_bias.argtypes = [p, p, p, p, i, i, i, i, p, p, i, p]
    lib.transoar_gemm_abi_version.restype = i
    if lib.transoar_gemm_abi_version() != ABI_VERSION:
        raise _native.NativeLibraryError("%s: ABI mismatch, rebuild" % _LIB_PATH)
    return lib


lib = _load()


def usable(x, w):
    """2-D row-major 16-bit operands the kernel takes as they are."""
    return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and w.dtype == x.dtype and x.dim() == 2 and w.dim() == 2
            and x.stride(1) == 1 and w.stride(1) == 1 and x.shape[1] == w.shape[1] and x.shape[1] % 8 == 0
            and w.shape[0] % 4 == 0 and x.stride(0) % 8 == 0 and w.stride(0) % 8 == 0
            and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0 and x.shape[0] * x.stride(0) * 2 < 0x7ffffff0)


def stream_kind(x, w, out_dtype=None, relu=False):
    """Which streaming kernel takes this product (None: the generic tiled kernel): bf16, dense operands, K = 384 with
    N % 64 == 0, or N = 384 with K % 32 == 0 (no ReLU there), tall enough to fill the chip."""
    if not STREAM or x.dtype != torch.bfloat16 or (out_dtype or x.dtype) != torch.bfloat16:
        return None
    m, k = x.shape
    n = w.shape[0]
    if not (x.is_contiguous() and w.is_contiguous() and m >= 16384):
        return None
    if k == 384 and n % 64 == 0 and n * 768 < 0x7ffffff0 and (n != 384 or STREAM_SQUARE) and (n <= 2048 or relu):
        # (384 x 384: the tiled kernel measures 0.149 ms against 0.158, profiles/r04_gemm_bench.jsonl; 384 -> 3072, the FPN's
        # transposed convolution as a product: 0.77 against 0.83, profiles/r05_gemm_bench.jsonl)
        return "k384"
    if STREAM_N384 and n == 384 and k % 32 == 0 and k != 384 and (m + 128) * k * 2 < 0xffffffff:
        return "n384"            # (off by default: 0.315 ms on 234 000 x 1024 -> 384 against the tiled kernel's 0.269)
    return None


def k384_takes(m, n):
    """Would `stream_kind` give the K = 384 streaming kernel an (m, 384) x (n, 384) product of dense, aligned bf16
    operands?  The shape part of its answer, for callers that decide before the bf16 operands exist; `gated` products
    (linear_gate) also read an (m, n) bf16 gate through a 32-bit byte offset."""
    return bool(STREAM and m >= 16384 and n % 64 == 0 and n * 768 < 0x7ffffff0 and (n != 384 or STREAM_SQUARE)
                and (m + 32) * n * 2 < 0xffffffff)          # (callers with fused epilogues: n = 1024 in the shipped configurations)


def linear_nt(x, w, bias=None, relu=False, out_dtype=None):
    """x (M, K), w (N, K) -> (M, N) = x @ w.T (+ bias) (ReLU); bias fp32 (N,)."""
    if not usable(x, w):
        raise RuntimeError("transoar_gemm_nt: operands must be 2-D row-major bf16/f16 CUDA tensors with K % 8 == 0, N % 4 == 0")
    m, k = x.shape
    n = w.shape[0]
    out_dtype = out_dtype or x.dtype
    out = torch.empty((m, n), dtype=out_dtype, device=x.device)
    if bias is not None:
        bias = bias.float().contiguous()
    kind = stream_kind(x, w, out_dtype)
    if kind == "n384" and relu:
        kind = None
    if kind is not None:
        with torch.cuda.device(x.device):
            bp = None if bias is None else bias.data_ptr()
            if kind == "k384":
                rc = lib.transoar_gemm_k384(x.data_ptr(), w.data_ptr(), bp, out.data_ptr(), m, n, 1 if relu else 0,
                                            torch.cuda.current_stream().cuda_stream)
            else:
                rc = lib.transoar_gemm_n384(x.data_ptr(), w.data_ptr(), bp, out.data_ptr(), m, k, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("transoar_gemm_%s failed with code %d" % (kind, rc))
        return out
    with torch.cuda.device(x.device):
        rc = lib.transoar_gemm_nt(x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(), out.data_ptr(),
                                  m, n, k, x.stride(0), w.stride(0), n, _DT[x.dtype], _DT[out_dtype], 1 if relu else 0,
                                  torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("transoar_gemm_nt failed with code %d" % rc)
    return out


def gelu_usable(x, w):
    """bf16 operands of `linear_nt` whose (M, N) result the GELU epilogues can address (byte offsets fit 31 bits)."""
    return usable(x, w) and x.dtype == torch.bfloat16 and x.shape[0] * w.shape[0] * 2 < 0x7ffffff0


def linear_gelu(x, w, bias=None):
    """-> (h, gelu(h)), h = x @ w.T + bias, both bf16 (M, N): fc1 and the activation of an MLP in one kernel
    (transoar_gemm_nt_gelu, TRANSOAR_GEMM_GELU_FORWARD)."""
    if not gelu_usable(x, w):
        raise RuntimeError("transoar_gemm_nt_gelu: operands must be 2-D row-major bf16 CUDA tensors with K % 8 == 0, N % 4 == 0")
    m, k = x.shape
    n = w.shape[0]
    h = torch.empty((m, n), dtype=torch.bfloat16, device=x.device)
    a = torch.empty_like(h)
    if bias is not None:
        bias = bias.float().contiguous()
    with torch.cuda.device(x.device):
        rc = lib.transoar_gemm_nt_gelu(x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(), a.data_ptr(), h.data_ptr(),
                                       m, n, k, x.stride(0), w.stride(0), n, 1, torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("transoar_gemm_nt_gelu failed with code %d" % rc)
    return h, a


def linear_gelu_grad(gy, wt, h):
    """-> (gy @ wt.T) * gelu'(h), bf16 (M, N): the data gradient of the layer AFTER a GELU with the activation's backward in the
    epilogue (TRANSOAR_GEMM_GELU_BACKWARD).  gy (M, K), wt (N, K) = that layer's weight transposed, h (M, N) the GELU's input."""
    if not gelu_usable(gy, wt) or h.dtype != torch.bfloat16 or not h.is_contiguous() or tuple(h.shape) != (gy.shape[0], wt.shape[0]):
        raise RuntimeError("transoar_gemm_nt_gelu: bf16 row-major operands, h (M, N) contiguous")
    m, k = gy.shape
    n = wt.shape[0]
    out = torch.empty((m, n), dtype=torch.bfloat16, device=gy.device)
    with torch.cuda.device(gy.device):
        rc = lib.transoar_gemm_nt_gelu(gy.data_ptr(), wt.data_ptr(), None, out.data_ptr(), h.data_ptr(), m, n, k, gy.stride(0),
                                       wt.stride(0), n, 2, torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("transoar_gemm_nt_gelu failed with code %d" % rc)
    return out


def linear_relu_dropout(x, w, bias, seed, keep_prob):
    """dropout(relu(x @ w.T + bias)) in one kernel (K = 384 streaming GEMM with the seeded mask of tokens.relu_dropout in
    its epilogue).  x (M, 384), w (N, 384) bf16 dense, bias fp32 or None, seed: device int32 tensor (tokens.dropout_seed)
    or None (no dropout)."""
    if stream_kind(x, w, relu=True) != "k384":
        raise RuntimeError("linear_relu_dropout: needs the K = 384 streaming kernel (dense bf16 operands, >= 16384 rows)")
    m, n = x.shape[0], w.shape[0]
    out = torch.empty((m, n), dtype=torch.bfloat16, device=x.device)
    if bias is not None:
        bias = bias.float().contiguous()
    with torch.cuda.device(x.device):
        rc = lib.transoar_gemm_k384_drop(x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(), out.data_ptr(), m, n, 1,
                                         None if seed is None else seed.data_ptr(), float(keep_prob), 1.0 / float(keep_prob),
                                         torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("transoar_gemm_k384_drop failed with code %d" % rc)
    return out


WGRAD384 = os.environ.get("TRANSOAR_GEMM_WGRAD384", "1") != "0"


def wgrad384_shapes(gy, x):
    """What the kernel can take at all: dense bf16 (T, n) / (T, k), 384 channels on one side, a multiple of 128 on the other."""
    if not (gy.is_cuda and gy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and gy.dim() == 2 and x.dim() == 2
            and gy.is_contiguous() and x.is_contiguous() and gy.shape[0] == x.shape[0]):
        return 0
    n, k = gy.shape[1], x.shape[1]
    other = n if k == 384 else (k if n == 384 else 0)
    return other if (other > 0 and other % 128 == 0 and gy.shape[0] * max(other, 384) * 2 < 0x7ffffff0) else 0


def wgrad384_usable(gy, x):
    """Where the token-streaming weight gradient (csrc/gemm_stream.hip, wgrad384_kernel) is the faster one: 256-wide
    column tiles (the other side a multiple of 256) and at most four of them -- every workgroup streams the whole
    384-channel operand through LDS-DMA, whose chip-wide rate (~6 TB/s, MI355X_MICROARCH.md "ldsdma-fill") is what bounds
    the kernel: 234 000 x 1024 x 384 in 0.21 ms against the one-tap conv GEMM's 0.29, but 384 x 384 (128-wide tiles)
    0.143 against 0.109 and 384 x 3072 0.83 against 0.73 (profiles/r04_gemm_bench.jsonl)."""
    other = wgrad384_shapes(gy, x)
    return WGRAD384 and other % 256 == 0 and 0 < other <= 1024 and gy.shape[0] >= 16384


def wgrad384(gy, x, with_bias=False):
    """gy (T, n), x (T, k) -> dW (n, k) fp32 = gy^T x;  with_bias: (dW, db) with db (n,) fp32 = gy.sum(0), summed from the
    fragments of the same pass (the bias gradient of the layer: no second read of gy)."""
    t, n = gy.shape
    k = x.shape[1]
    if k == 384:
        a, b, na, tr, side = gy, x, n, 0, 1
    else:
        a, b, na, tr, side = x, gy, k, 1, 2
    chunks = lib.transoar_gemm_wgrad384_chunks(t, na)
    part = torch.empty((chunks, n, k), dtype=torch.float32, device=gy.device)
    out = torch.empty((n, k), dtype=torch.float32, device=gy.device)
    bias_part = torch.empty((chunks, n), dtype=torch.float32, device=gy.device) if with_bias else None
    db = torch.empty((n,), dtype=torch.float32, device=gy.device) if with_bias else None
    with torch.cuda.device(gy.device):
        rc = lib.transoar_gemm_wgrad384_bias(a.data_ptr(), b.data_ptr(), part.data_ptr(), out.data_ptr(), t, na, tr, chunks,
                                             bias_part.data_ptr() if with_bias else None, db.data_ptr() if with_bias else None,
                                             side if with_bias else 0, torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("transoar_gemm_wgrad384 failed with code %d" % rc)
    return (out, db) if with_bias else out


def linear_gate(x, w, gate, scale):
    """gate > 0 ? (x @ w.T) * scale : 0 in one kernel: x (M, 384), w (N, 384), gate (M, N) bf16 dense -- the data gradient of
    the FFN's second layer with the gradient of dropout(relu(.)) in the GEMM's epilogue (gate = the saved hidden tensor)."""
    if stream_kind(x, w, relu=True) != "k384" or gate.shape != (x.shape[0], w.shape[0]) or gate.dtype != torch.bfloat16 or not gate.is_contiguous():
        raise RuntimeError("linear_gate: needs the K = 384 streaming kernel (dense bf16 operands, >= 16384 rows) and a dense bf16 gate")
    m, n = x.shape[0], w.shape[0]
    if (m + 32) * n * 2 >= 0xffffffff:
        raise RuntimeError("linear_gate: the gate is addressed with 32-bit byte offsets; chunk over the rows")
    out = torch.empty((m, n), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.transoar_gemm_k384_gate(x.data_ptr(), w.data_ptr(), gate.data_ptr(), out.data_ptr(), m, n, float(scale),
                                         torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("transoar_gemm_k384_gate failed with code %d" % rc)
    return out
