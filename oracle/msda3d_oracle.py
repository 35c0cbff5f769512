"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the scalar C oracle.

``libmsda3d_oracle.so`` (built by ``make -C oracle``) restates the reference's
CUDA kernels sequentially on the CPU:
  forward   ops/src/cuda/ms_deform_im2col_cuda.cuh:370-439, :31-114
  backward  ops/src/cuda/ms_deform_im2col_cuda.cuh:116-241 (+ the channel
            reduction of :551-661)
This module only marshals numpy arrays into it.  float32 and float64 builds
are exposed; the float build does all arithmetic in float like a float kernel.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmsda3d_oracle.so")
_lib = None


def build(force=False):
    """Compile the C oracle with gcc (seconds).  Idempotent."""
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(value, shapes, lsi, loc, attn):
    dt = value.dtype
    if dt not in (np.float32, np.float64):
        raise TypeError("oracle handles float32/float64 only, got %s" % dt)
    value = np.ascontiguousarray(value)
    loc = np.ascontiguousarray(loc, dtype=dt)
    attn = np.ascontiguousarray(attn, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    lsi = np.ascontiguousarray(lsi, dtype=np.int64)
    N, S, M, C = value.shape
    _, Lq, M2, L, P, three = loc.shape
    assert M2 == M and three == 3 and shapes.shape == (L, 3) and lsi.shape == (L,)
    assert attn.shape == (N, Lq, M, L, P)
    assert int((shapes[:, 0] * shapes[:, 1] * shapes[:, 2]).sum()) == S
    dims = [ctypes.c_int(int(x)) for x in (N, S, M, C, L, Lq, P)]
    sfx = "f32" if dt == np.float32 else "f64"
    return value, shapes, lsi, loc, attn, dims, sfx, (N, S, M, C, L, Lq, P)


def forward(value, shapes, lsi, loc, attn):
    """-> out (N, Lq, M*C), same dtype as ``value``."""
    value, shapes, lsi, loc, attn, dims, sfx, (N, S, M, C, L, Lq, P) = _prep(
        value, shapes, lsi, loc, attn)
    out = np.empty((N, Lq, M * C), dtype=value.dtype)
    fn = getattr(_load(), "msda3d_oracle_forward_" + sfx)
    fn.restype = None
    fn(_ptr(value), _ptr(shapes), _ptr(lsi), _ptr(loc), _ptr(attn), _ptr(out), *dims)
    return out


def backward(value, shapes, lsi, loc, attn, grad_out):
    """-> (grad_value, grad_loc, grad_attn) with the input shapes."""
    value, shapes, lsi, loc, attn, dims, sfx, _ = _prep(value, shapes, lsi, loc, attn)
    grad_out = np.ascontiguousarray(grad_out, dtype=value.dtype)
    gv = np.empty_like(value)
    gl = np.empty_like(loc)
    ga = np.empty_like(attn)
    fn = getattr(_load(), "msda3d_oracle_backward_" + sfx)
    fn.restype = None
    fn(_ptr(value), _ptr(shapes), _ptr(lsi), _ptr(loc), _ptr(attn), _ptr(grad_out),
       _ptr(gv), _ptr(gl), _ptr(ga), *dims)
    return gv, gl, ga
