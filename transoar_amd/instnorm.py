"""Autograd shim of the fused InstanceNorm3d(affine)+ReLU kernels
(include/transoar_instnorm.h) for channels-last bf16 activations."""
import ctypes
import os

import torch

from . import _native  # noqa: F401
from .conv3d import to_ndhwc

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_PKG, "libtransoar_instnorm.so")


def _load():
    if not os.path.exists(_LIB_PATH):
        raise _native.NativeLibraryError("%s is not built (python transoar_amd/_build.py)" % _LIB_PATH)
    lib = ctypes.CDLL(_LIB_PATH)
    i, p, lg, f = ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_float
    lib.transoar_instnorm_relu_forward.restype = i
    lib.transoar_instnorm_relu_forward.argtypes = [p, p, p, p, p, p, i, lg, i, f, i, p]
    lib.transoar_instnorm_relu_forward_parts.restype = i
    lib.transoar_instnorm_relu_forward_parts.argtypes = [p, p, p, p, p, i, p, p, i, lg, i, f, i, p]
    lib.transoar_instnorm_relu_backward.restype = i
    lib.transoar_instnorm_relu_backward.argtypes = [p, p, p, p, p, p, p, p, i, lg, i, i, p]
    lib.transoar_instnorm_abi_version.restype = i
    if lib.transoar_instnorm_abi_version() != 3:
        raise _native.NativeLibraryError("%s: ABI mismatch, rebuild" % _LIB_PATH)
    return lib


lib = _load()
CL3D = torch.channels_last_3d


def supported(x, channels):
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 5 and channels % 8 == 0
            and 192 % (channels // 8) == 0)


class _InstNormReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, part=None):
        x = to_ndhwc(x)
        n, c = x.shape[:2]
        v = x.shape[2] * x.shape[3] * x.shape[4]
        y = torch.empty_like(x, memory_format=CL3D)
        g32, b32 = gamma.float().contiguous(), beta.float().contiguous()
        ws = torch.empty((n, c, 2), dtype=torch.float64, device=x.device)
        mean_rstd = torch.empty((n, c, 2), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            if part is not None:
                # the statistics were taken in the epilogue of the convolution that produced x (conv3d.Conv3dK3.forward_with_stats)
                rc = lib.transoar_instnorm_relu_forward_parts(x.data_ptr(), g32.data_ptr(), b32.data_ptr(), y.data_ptr(),
                                                              part.data_ptr(), part.shape[0] // n, ws.data_ptr(), mean_rstd.data_ptr(),
                                                              n, v, c, float(eps), 1 if relu else 0,
                                                              torch.cuda.current_stream().cuda_stream)
            else:
                rc = lib.transoar_instnorm_relu_forward(x.data_ptr(), g32.data_ptr(), b32.data_ptr(), y.data_ptr(),
                                                        ws.data_ptr(), mean_rstd.data_ptr(), n, v, c, float(eps),
                                                        1 if relu else 0, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("transoar_instnorm_relu_forward failed with code %d" % rc)
        ctx.save_for_backward(x, g32, b32, mean_rstd)
        ctx.relu, ctx.param_dtype = relu, gamma.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g32, b32, mean_rstd = ctx.saved_tensors
        dy = to_ndhwc(dy)
        n, c = x.shape[:2]
        v = x.shape[2] * x.shape[3] * x.shape[4]
        dx = torch.empty_like(x, memory_format=CL3D)
        red = torch.empty((n, c, 2), dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            dparams = torch.empty((2, c), dtype=torch.float32, device=x.device)      # dbeta | dgamma, summed over the samples by the kernel
            rc = lib.transoar_instnorm_relu_backward(x.data_ptr(), dy.data_ptr(), g32.data_ptr(), b32.data_ptr(),
                                                     mean_rstd.data_ptr(), dx.data_ptr(), red.data_ptr(), dparams.data_ptr(), n, v, c,
                                                     1 if ctx.relu else 0, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("transoar_instnorm_relu_backward failed with code %d" % rc)
        return dx, dparams[1].to(ctx.param_dtype), dparams[0].to(ctx.param_dtype), None, None, None


def instance_norm_relu(x, gamma, beta, eps=1e-5, relu=True, part=None):
    """relu(instance_norm(x) * gamma + beta) for (N,C,D,H,W) bf16 on the GPU.  part: the per-workgroup partial sums of x's
    statistics when the producing convolution took them in its epilogue (conv3d.Conv3dK3.forward_with_stats)."""
    return _InstNormReLU.apply(x, gamma, beta, eps, relu, part)
