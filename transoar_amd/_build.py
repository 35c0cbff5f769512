"""In-tree hipcc build of the gfx950 shared libraries (no torch cpp_extension,
no hipify: the sources are HIP written for CDNA4 and compiled as-is).

    python transoar_amd/_build.py [--force]   # run by PATH: importing the package
                                              # would load the (possibly stale) .so
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]

# library name -> sources (relative to csrc/)
LIBS = {
    "libtransoar_msda3d.so": ["msda3d.hip"],
    "libtransoar_conv3d.so": ["conv3d.hip"],
    "libtransoar_instnorm.so": ["instnorm.hip"],
    "libtransoar_rows.so": ["rows.hip"],
    "libtransoar_tokens.so": ["tokens.hip"],
    "libtransoar_gemm.so": ["gemm.hip"],
    "libtransoar_convgemm.so": ["conv_gemm.hip"],
}


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(PKG, "..", "include", f) for f in os.listdir(os.path.join(PKG, "..", "include"))]


def lib_path(name):
    return os.path.join(PKG, name)


def is_stale(name):
    out = lib_path(name)
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=True):
    """Compile every stale library.  Cross-compiles without a GPU."""
    built = []
    for name, srcs in LIBS.items():
        if not (force or is_stale(name)):
            continue
        cmd = [HIPCC] + COMMON_FLAGS + [os.path.join(CSRC, s) for s in srcs] + ["-o", lib_path(name)]
        if verbose:
            print("[transoar_amd] " + " ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        built.append(name)
    return built


if __name__ == "__main__":
    build(force="--force" in sys.argv)
