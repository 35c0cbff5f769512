"""In-tree hipcc build of the gfx950 shared libraries (no torch cpp_extension,
no hipify: the sources are HIP written for CDNA4 and compiled as-is).

    python transoar_amd/_build.py [--force]   # run by PATH: importing the package
                                              # would load the (possibly stale) .so

A library is rebuilt when one of the files it includes (hipcc's own depfile of
the last build, kept next to the .so as `.<name>.d`) is newer than it; the
stale libraries are compiled in parallel.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]

# library name -> sources (relative to csrc/)
LIBS = {
    "libtransoar_msda3d.so": ["msda3d.hip", "msda3d_sort.hip"],
    "libtransoar_conv3d.so": ["conv3d.hip"],
    "libtransoar_instnorm.so": ["instnorm.hip"],
    "libtransoar_rows.so": ["rows.hip"],
    "libtransoar_tokens.so": ["tokens.hip"],
    "libtransoar_gemm.so": ["gemm.hip", "gemm_stream.hip"],
    "libtransoar_convgemm.so": ["conv_gemm.hip"],
    "libtransoar_attn.so": ["attn.hip"],
    "libtransoar_optim.so": ["optim.hip"],
    "libtransoar_criterion.so": ["criterion.hip"],
}


def lib_path(name):
    return os.path.join(PKG, name)


def _dep_path(name):
    return os.path.join(PKG, "." + name + ".d")


def _deps(name):
    """Files the library was built from: the depfile of its last build (project files only), or, without
    one, every file of csrc/ and include/."""
    try:
        if len(LIBS[name]) != 1:
            raise OSError("no depfile for a library of several sources")
        with open(_dep_path(name)) as f:
            words = f.read().replace("\\\n", " ").split()
        deps = [w for w in words[1:] if not w.startswith("/opt/") and not w.startswith("/usr/")]
        if deps:
            return deps + [os.path.join(CSRC, s) for s in LIBS[name]]
    except OSError:
        pass
    inc = os.path.join(PKG, "..", "include")
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(inc, f) for f in os.listdir(inc)]


def is_stale(name):
    out = lib_path(name)
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    for d in _deps(name):
        try:
            if os.path.getmtime(d) > t:
                return True
        except OSError:          # a file of the last build is gone
            return True
    return False


def _compile(name, verbose):
    cmd = [HIPCC] + COMMON_FLAGS + [os.path.join(CSRC, s) for s in LIBS[name]] + ["-o", lib_path(name)]
    if len(LIBS[name]) == 1:
        cmd += ["-MD", "-MF", _dep_path(name)]
    elif os.path.exists(_dep_path(name)):
        os.remove(_dep_path(name))       # a depfile from when the library had one source: it misses the newer headers
    if verbose:
        print("[transoar_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return name


def build(force=False, verbose=True, only=None):
    """Compile every stale library.  Cross-compiles without a GPU."""
    names = [n for n in LIBS if (only is None or n in only) and (force or is_stale(n))]
    if not names:
        return []
    with ThreadPoolExecutor(max_workers=min(len(names), os.cpu_count() or 4)) as pool:
        return list(pool.map(lambda n: _compile(n, verbose), names))


if __name__ == "__main__":
    build(force="--force" in sys.argv)
