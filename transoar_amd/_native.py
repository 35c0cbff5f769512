"""ctypes binding of the C ABI declared in include/transoar_msda3d.h.

There is deliberately NO fallback: if the gfx950 library is missing or does
not export the ABI, importing this module raises.  A missing build must never
turn into a silently slower (or CPU) path.
"""
import ctypes
import os

# torch bundles its own libamdhip64.so.7 / libhsa-runtime64; it has to be in the
# process BEFORE this library is dlopen'ed, so that both resolve to the same
# HIP runtime (loading /opt/rocm's copy first leaves torch and the kernels on
# two different runtimes: "no ROCm-capable device is detected").
import torch  # noqa: F401

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libtransoar_msda3d.so"

F32, F64, BF16, F16 = 0, 1, 2, 3
FORCE_GENERIC = 1
ABI_VERSION = 6


class NativeLibraryError(ImportError):
    pass


def _load():
    path = os.path.join(_PKG, _LIB_NAME)
    if not os.path.exists(path):
        raise NativeLibraryError(
            "%s is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `python transoar_amd/_build.py`) in the repository root; there is no "
            "fallback implementation." % path)
    lib = ctypes.CDLL(path)
    c_int, c_void_p, c_uint = ctypes.c_int, ctypes.c_void_p, ctypes.c_uint
    try:
        lib.transoar_msda3d_forward.restype = c_int
        lib.transoar_msda3d_forward.argtypes = [c_void_p] * 6 + [c_int] * 9 + [c_void_p, c_uint, c_void_p]
        lib.transoar_msda3d_forward_fused.restype = c_int
        lib.transoar_msda3d_forward_fused.argtypes = ([c_void_p] * 3 + [ctypes.c_long, c_void_p] + [c_int] * 7 +
                                                      [c_void_p, c_void_p])
        lib.transoar_msda3d_backward.restype = c_int
        lib.transoar_msda3d_backward.argtypes = ([c_void_p] * 10 + [ctypes.c_size_t] + [c_int] * 9 +
                                                 [c_void_p, c_uint, c_void_p])
        lib.transoar_msda3d_backward_proj.restype = c_int
        lib.transoar_msda3d_backward_proj.argtypes = ([c_void_p] * 9 + [ctypes.c_size_t] + [c_int] * 9 +
                                                      [c_void_p, c_uint, c_void_p])
        lib.transoar_msda3d_backward_workspace_bytes.restype = ctypes.c_size_t
        lib.transoar_msda3d_backward_workspace_bytes.argtypes = [c_int] * 9 + [c_uint]
        lib.transoar_msda3d_profile_enable.restype = None
        lib.transoar_msda3d_profile_enable.argtypes = [c_int]
        lib.transoar_msda3d_profile_read.restype = c_int
        lib.transoar_msda3d_profile_read.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)]
        lib.transoar_msda3d_strerror.restype = ctypes.c_char_p
        lib.transoar_msda3d_strerror.argtypes = [c_int]
        lib.transoar_msda3d_abi_version.restype = c_int
        lib.transoar_msda3d_abi_version.argtypes = []
    except AttributeError as e:  # symbol missing
        raise NativeLibraryError("%s does not export the transoar_msda3d ABI: %s" % (path, e))
    got = lib.transoar_msda3d_abi_version()
    if got != ABI_VERSION:
        raise NativeLibraryError("%s has ABI version %d, expected %d: rebuild" % (path, got, ABI_VERSION))
    return lib


lib = _load()
LIB_PATH = os.path.join(_PKG, _LIB_NAME)


def check(code, what):
    if code != 0:
        raise RuntimeError("%s failed: %s (code %d)" % (
            what, lib.transoar_msda3d_strerror(code).decode(), code))


PROF_KINDS = ("fwd", "bwd_query", "cell_count", "scan", "cell_fill", "pull", "fwd_generic", "bwd_generic",
              "value_tile", "value_cells")


def profile_enable(on):
    lib.transoar_msda3d_profile_enable(1 if on else 0)


def profile_read():
    """-> {kind: (total_ms, launches)} for the kernels recorded since the last read."""
    ms = (ctypes.c_double * len(PROF_KINDS))()
    n = (ctypes.c_long * len(PROF_KINDS))()
    check(lib.transoar_msda3d_profile_read(ms, n), "transoar_msda3d_profile_read")
    return {k: (ms[i], n[i]) for i, k in enumerate(PROF_KINDS)}
