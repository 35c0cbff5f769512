"""CPU: the C-ABI shared library builds/loads, exports every symbol the
header declares, and validates arguments on the host (no kernel is launched
here -- there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "transoar_msda3d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(transoar_msda3d_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = _declared_symbols()
    assert "transoar_msda3d_forward" in syms and "transoar_msda3d_backward" in syms


def test_library_exports_every_declared_symbol():
    from transoar_amd import _native
    lib = ctypes.CDLL(_native.LIB_PATH)
    for s in _declared_symbols():
        assert hasattr(lib, s), s
    assert _native.lib.transoar_msda3d_abi_version() == _native.ABI_VERSION


def test_every_header_under_include_has_its_library_and_symbols():
    """include/transoar_<name>.h  <->  transoar_amd/libtransoar_<name>.so: every function a header declares is
    exported by the library of the same name."""
    inc = os.path.join(ROOT, "include")
    seen = 0
    for fn in sorted(os.listdir(inc)):
        m = re.match(r"transoar_([a-z0-9]+)\.h$", fn)
        if not m:
            continue
        text = re.sub(r"/\*.*?\*/", "", open(os.path.join(inc, fn)).read(), flags=re.S)
        syms = sorted(set(re.findall(r"\b(transoar_[a-z_0-9]+)\s*\(", text)))
        lib = ctypes.CDLL(os.path.join(ROOT, "transoar_amd", "libtransoar_%s.so" % m.group(1)))
        assert syms, fn
        for s in syms:
            assert hasattr(lib, s), (fn, s)
        seen += 1
    assert seen >= 6


def test_gemm_argument_errors():
    lib = ctypes.CDLL(os.path.join(ROOT, "transoar_amd", "libtransoar_gemm.so"))
    buf = (ctypes.c_char * 64)()
    p16 = (ctypes.addressof(buf) + 15) & ~15
    f = lib.transoar_gemm_nt
    f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 9 + [ctypes.c_void_p]
    assert f(None, p16, None, p16, 8, 8, 8, 8, 8, 8, 2, 2, 0, None) == -1
    assert f(p16, p16, None, p16, 8, 8, 12, 12, 12, 8, 2, 2, 0, None) == -2       # K not a multiple of 8
    assert f(p16, p16, None, p16, 8, 8, 8, 8, 8, 8, 0, 0, 0, None) == -3          # fp32 operands
    assert f(p16 + 2, p16, None, p16, 8, 8, 8, 8, 8, 8, 2, 2, 0, None) == -4


def test_argument_errors_are_returned_not_printed():
    from transoar_amd import _native
    lib = _native.lib
    buf = (ctypes.c_char * 64)()
    p = ctypes.addressof(buf)
    p16 = (p + 15) & ~15
    dims_ok = (1, 8, 1, 4, 1, 1, 1)
    # NULL pointer
    assert lib.transoar_msda3d_forward(None, p16, p16, p16, p16, p16, *dims_ok, 0, 0, None, 0, None) == -1
    # bad dimension
    assert lib.transoar_msda3d_forward(p16, p16, p16, p16, p16, p16, 1, 8, 1, 0, 1, 1, 1, 0, 0, None, 0, None) == -2
    # dtype combination: f32 value with f64 loc
    assert lib.transoar_msda3d_forward(p16, p16, p16, p16, p16, p16, *dims_ok, 0, 1, None, 0, None) == -3
    # bf16 value with fp32 loc is legal as far as dtype goes; misaligned buffer
    assert lib.transoar_msda3d_forward(p16 + 4, p16, p16, p16, p16, p16, *dims_ok, 2, 0, None, 0, None) == -4
    # too many levels
    assert lib.transoar_msda3d_forward(p16, p16, p16, p16, p16, p16, 1, 8, 1, 4, 9, 1, 1, 0, 0, None, 0, None) == -5
    assert lib.transoar_msda3d_backward(p16, p16, p16, p16, p16, None, p16, p16, p16, None, 0, *dims_ok, 0, 0, None, 0, None) == -1
    assert lib.transoar_msda3d_backward(p16, p16, p16, p16, p16, p16, p16, p16, p16, None, 0, 2, 1000, 6, 64, 1, 10, 4, 0, 0, None, 0, None) == -6
    assert lib.transoar_msda3d_backward_workspace_bytes(2, 1000, 6, 64, 1, 10, 4, 0, 0, 0) > 0
    for code in (0, -1, -2, -3, -4, -5, -6, -7):
        assert len(lib.transoar_msda3d_strerror(code)) > 0


def test_cpu_tensors_raise_like_the_reference():
    """ops/src/ms_deform_attn.h:38: "Not implemented on the CPU"."""
    import torch
    from transoar_amd import MSDA
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(torch.zeros(1, 8, 1, 4), torch.tensor([[2, 2, 2]]), torch.tensor([0]),
                                    torch.zeros(1, 1, 1, 1, 1, 3), torch.zeros(1, 1, 1, 1, 1), 64)


def test_use_cuda_false_needs_an_injected_core():
    import torch
    from transoar_amd import MSDeformAttn
    from transoar_amd import ms_deform_attn as mod
    prev = mod.register_debug_core(None)
    try:
        m = MSDeformAttn(12, 1, 6, 1, use_cuda=False)
        with pytest.raises(RuntimeError, match="no CPU path"):
            m(torch.zeros(1, 8, 12), torch.zeros(1, 8, 1, 3), torch.zeros(1, 8, 12),
              torch.tensor([[2, 2, 2]]), torch.tensor([0]))
    finally:
        mod.register_debug_core(prev)


def test_sampling_head_and_colsum_argument_errors():
    tok = ctypes.CDLL(os.path.join(ROOT, "transoar_amd", "libtransoar_tokens.so"))
    rows = ctypes.CDLL(os.path.join(ROOT, "transoar_amd", "libtransoar_rows.so"))
    buf = (ctypes.c_char * 64)()
    p16 = (ctypes.addressof(buf) + 15) & ~15
    p, lg, i = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
    f = tok.transoar_sampling_head_forward
    f.argtypes = [p, p, lg, p, p, p, lg, i, i, i, p]
    assert f(None, p16, 4, p16, p16, p16, 4, 6, 4, 4, None) == -1
    assert f(p16, p16, 4, p16, p16, p16, 4, 6, 65, 4, None) == -2           # L * P > 256
    assert f(p16, p16, 3, p16, p16, p16, 4, 6, 4, 4, None) == -2            # tokens not a multiple of ref_rows
    b = tok.transoar_sampling_head_backward
    b.argtypes = [p, p, p, p, p, lg, i, i, i, p]
    assert b(p16, p16, None, p16, p16, 4, 6, 4, 4, None) == -1
    q = tok.transoar_pos_query_forward
    q.argtypes = [p, p, p, p, i, lg, p, lg, i, p]
    assert q(None, p16, p16, p16, 4, 8, p16, 16, 384, None) == -1
    assert q(p16, p16, p16, p16, 4, 8, p16, 16, 200, None) == -2           # cols not a multiple of 128
    assert q(p16, p16, p16, p16, 9, 8, p16, 16, 384, None) == -3           # more than TRANSOAR_TOK_MAX_LEVELS
    qb = tok.transoar_pos_query_backward
    qb.argtypes = [p, p, i, lg, p, lg, i, p]
    assert qb(p16, p16, 4, 8, None, 16, 384, None) == -1
    assert tok.transoar_pos_query_partial_rows() > 0 and tok.transoar_tokens_abi_version() == 7
    c = rows.transoar_rows_colsum
    c.argtypes = [p, p, p, lg, i, p]
    assert c(None, p16, p16, 8, 8, None) == -1
    assert c(p16, p16, p16, 8, 12, None) == -2                             # cols not a multiple of 8
    assert rows.transoar_rows_colsum_workspace_floats(384) == 1024 * 384
    cs = rows.transoar_rows_colsum_small
    cs.argtypes = [p, p, lg, i, i, p]
    assert cs(None, p16, 8, 8, 1, None) == -1
    assert cs(p16, p16, 8, 12, 1, None) == -2                              # bf16: cols not a multiple of 8
    assert cs(p16, p16, 8, 6, 0, None) == -2                               # fp32: cols not a multiple of 4
    # round 5: the LayerNorm-rows backward takes the shortcut's gradient (NULL = none); the merge gather with its residual
    lb = tok.transoar_ln_rows_backward
    lb.argtypes = [p, p, i, p, p, p, p, p, p, lg, i, p]
    assert lb(p16, p16, 1, p16, p16, p16, None, None, p16, 8, 48, None) == -1      # dx is required, dx_add is not
    assert lb(p16, p16, 1, p16, p16, p16, None, p16, p16, 8, 52, None) == -2       # cols not a multiple of 8
    ga = rows.transoar_rows_gather_axpy
    ga.argtypes = [p, p, p, p, p, i, lg, lg, i, p]
    assert ga(None, p16, None, None, p16, 1, 8, 8, 96, None) == -1
    assert ga(p16, p16, None, None, p16, 1, 8, 8, 100, None) == -2                 # rows not a multiple of 16 bytes


def test_round5_gemm_entries_argument_errors():
    """transoar_gemm_nt_gelu (include/transoar_gemm.h) and transoar_linear_wgrad_bias (include/transoar_convgemm.h)."""
    gemm = ctypes.CDLL(os.path.join(ROOT, "transoar_amd", "libtransoar_gemm.so"))
    conv = ctypes.CDLL(os.path.join(ROOT, "transoar_amd", "libtransoar_convgemm.so"))
    attn = ctypes.CDLL(os.path.join(ROOT, "transoar_amd", "libtransoar_attn.so"))
    buf = (ctypes.c_char * 64)()
    p16 = (ctypes.addressof(buf) + 15) & ~15
    p, i, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    g = gemm.transoar_gemm_nt_gelu
    g.argtypes = [p] * 5 + [i] * 7 + [p]
    assert g(p16, p16, None, p16, None, 8, 8, 8, 8, 8, 8, 1, None) == -1           # aux is required
    assert g(p16, p16, None, p16, p16, 8, 8, 12, 12, 12, 8, 1, None) == -2         # K not a multiple of 8
    assert g(p16, p16, None, p16, p16, 8, 8, 8, 8, 8, 8, 3, None) == -2            # unknown mode
    assert g(p16, p16, None, p16, p16 + 2, 8, 8, 8, 8, 8, 8, 2, None) == -4
    assert gemm.transoar_gemm_abi_version() == 4
    w = conv.transoar_linear_wgrad_bias
    w.argtypes = [p] * 6 + [i] * 4 + [p]
    assert w(p16, p16, p16, p16, None, p16, 1024, 48, 144, 1, None) == -1
    assert w(p16, p16, p16, p16, p16, p16, 1024, 128, 144, 1, None) == -2          # no padding column for the ones
    wf = attn.transoar_win_attn_forward
    wf.argtypes = [p] * 5 + [i] * 5 + [f32, p]
    assert wf(p16, p16, None, p16, p16, 1, 1, 125, 3, 16, 0.0, None) == -2         # scale must be positive
    assert wf(p16, p16, None, p16, p16, 1, 1, 125, 3, 24, 0.25, None) == -2        # head dimension 16 or 32


def test_convgemm_and_fused_gather_argument_errors():
    """The round-3 entry points validate on the host too: the implicit-GEMM convolutions (include/transoar_convgemm.h) and
    the fused head + gather (transoar_msda3d_forward_fused)."""
    cg = ctypes.CDLL(os.path.join(ROOT, "transoar_amd", "libtransoar_convgemm.so"))
    buf = (ctypes.c_char * 64)()
    p16 = (ctypes.addressof(buf) + 15) & ~15
    p, i, u = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint
    f = cg.transoar_conv3d_igemm
    f.argtypes = [p] * 5 + [i] * 17 + [u] * 3 + [i, i, p]
    taps = 3 | (0 << 2) | (0 << 4) | (1 << 6) | (1 << 8) | (2 << 10) | (2 << 12)         # (-1,0) (0,1) (+1,2)
    geom = (1, 4, 4, 8, 16, 16, 4, 4, 8, 1, 4, 4, 8, 1, 0, 0, 0)
    assert f(None, p16, None, p16, None, *geom, taps, taps, taps, 1, 0, None) == -1
    assert f(p16, p16, None, p16, None, 1, 4, 4, 8, 12, 16, 4, 4, 8, 1, 4, 4, 8, 1, 0, 0, 0, taps, taps, taps, 1, 0, None) == -2    # Cin % 8
    assert f(p16, p16, None, p16, None, *geom, 0, taps, taps, 1, 0, None) == -2           # empty tap list
    assert f(p16, p16, None, p16, None, *geom, taps, taps, taps, 2, 0, None) == -2        # split > 1 needs the partial maps
    w = cg.transoar_conv3d_wgrad
    w.argtypes = [p] * 4 + [i] * 10 + [u] * 3 + [i, i, p]
    assert w(p16, p16, None, p16, 1, 4, 4, 8, 16, 16, 4, 4, 8, 1, taps, taps, taps, 4, 27, None) == -1
    assert w(p16, p16, p16, p16, 1, 4, 4, 8, 16, 16, 4, 4, 8, 1, taps, taps, taps, 4, 5, None) == -2       # taps_out is 1 or 27
    h = cg.transoar_conv3d_dgrad_s2_halo
    h.argtypes = [p, p, p] + [i] * 9 + [p]
    assert h(p16, None, p16, 1, 4, 4, 8, 8, 8, 16, 24, 48, None) == -1
    assert h(p16, p16, p16, 1, 4, 4, 8, 8, 8, 16, 40, 48, None) == -2        # Cin <= 32
    assert h(p16, p16, p16, 1, 4, 4, 8, 8, 8, 16, 24, 96, None) == -2        # Cout 16 / 32 / 48
    assert h(p16, p16, p16, 1, 4, 4, 8, 8, 8, 14, 24, 48, None) == -2        # W = 2 OW or 2 OW - 1
    rg = cg.transoar_conv3d_wgrad_ring
    rg.argtypes = [p] * 4 + [i] * 11 + [p]
    assert rg(p16, p16, None, p16, 1, 4, 4, 64, 48, 16, 4, 4, 64, 1, 8, None) == -1
    assert rg(p16, p16, p16, p16, 1, 4, 4, 64, 96, 16, 4, 4, 64, 1, 8, None) == -2      # Cin <= 64
    assert rg(p16, p16, p16, p16, 1, 4, 4, 48, 48, 16, 4, 4, 48, 1, 8, None) == -2      # MW % 64
    assert rg(p16, p16, p16, p16, 1, 4, 4, 64, 16, 16, 4, 4, 64, 1, 8, None) == -2      # <= 32 x 32 channels: not this kernel
    assert rg(p16, p16, p16, p16, 1, 4, 4, 128, 48, 16, 4, 4, 64, 1, 8, None) == -2     # MW = (SW - 1) / stride + 1
    cg.transoar_conv3d_wgrad_part_floats.restype = ctypes.c_long
    assert cg.transoar_conv3d_wgrad_part_floats(96, 96, 10, 27) == 27 * 96 * 96 * 10

    from transoar_amd import _native
    shapes = (ctypes.c_int64 * 3)(2, 2, 2)
    g = _native.lib.transoar_msda3d_forward_fused
    assert g(None, p16, p16, 8, p16, 1, 8, 6, 64, 1, 4, 2, ctypes.addressof(shapes), None) == -1
    assert g(p16, p16, p16, 8, p16, 1, 8, 6, 64, 1, 4, 0, ctypes.addressof(shapes), None) == -3      # fp32 value: 16-bit storage only
    assert g(p16, p16, p16, 8, p16, 1, 8, 6, 32, 1, 4, 2, ctypes.addressof(shapes), None) == -2      # 32 channels per head: not this kernel's form
    assert g(p16, p16, p16, 5, p16, 1, 8, 6, 64, 1, 4, 2, ctypes.addressof(shapes), None) == -2      # reference rows: S or N * S


def test_roi_attention_argument_errors():
    """include/transoar_attn.h (fused masked cross-attention): host-side validation."""
    lib = ctypes.CDLL(os.path.join(ROOT, "transoar_amd", "libtransoar_attn.so"))
    buf = (ctypes.c_char * 64)()
    p16 = (ctypes.addressof(buf) + 15) & ~15
    p, i, lg, z = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_size_t
    lib.transoar_roi_attn_workspace_bytes.restype = z
    lib.transoar_roi_attn_workspace_bytes.argtypes = [i, i, i]
    assert lib.transoar_roi_attn_workspace_bytes(40, 216, 1) >= 40 * 216 * 4
    assert lib.transoar_roi_attn_workspace_bytes(40, 216, 3) >= 40 * 216 * (4 + 3 * 384 * 4 + 2 * 3 * 4)
    assert lib.transoar_roi_attn_workspace_bytes(0, 216, 1) == 0
    f = lib.transoar_roi_attn_forward
    f.argtypes = [p] * 8 + [z, i, i, i, lg, i, i, p]
    assert f(None, p16, p16, p16, p16, p16, p16, p16, 0, 1, 1, 8, 32, 384, 1, None) == -1
    assert f(p16, p16, p16, p16, p16, p16, p16, p16, 1 << 20, 1, 1, 8, 32, 256, 1, None) == -2       # C != 384
    assert f(p16, p16, p16, p16, p16, p16, p16, p16, 1 << 20, 3, 2, 8, 32, 384, 1, None) == -2       # G not a multiple of O
    assert f(p16, p16, p16, p16, p16, p16, p16, p16, 1 << 20, 1, 1, 600, 32, 384, 1, None) == -2     # more than 512 rows per group
    assert f(p16, p16, p16, p16, p16, p16, p16, p16, 0, 1, 1, 8, 32, 384, 2, None) == -3             # workspace too small
    b = lib.transoar_roi_attn_backward
    b.argtypes = [p] * 11 + [z, i, i, i, lg, i, i, p]
    assert b(p16, p16, p16, p16, p16, p16, p16, p16, None, p16, p16, 1 << 20, 1, 1, 8, 32, 384, 1, None) == -1
    assert b(p16, p16, p16, p16, p16, p16, p16, p16, p16, p16, p16, 16, 1, 1, 8, 32, 384, 1, None) == -3
    assert lib.transoar_attn_abi_version() == 2
