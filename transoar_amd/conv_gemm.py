"""ctypes binding and autograd shim of the LDS-tiled implicit-GEMM convolution kernels (C ABI:
include/transoar_convgemm.h): the 3x3x3 convolutions of the encoder stages from 48 channels up and of the FPN `out`
layers (encoder_blocks.py:28-51, attn_fpn.py:65-73,126), forward, data gradient (stride 1: one launch with the
mirrored tap list; stride 2: the eight parity classes of the dx voxels in one launch -- up to 32 input channels the
halo-tile kernel that writes dx as whole lines) and weight gradient -- and the
weight gradient of a token projection as the one-tap case.  No fallback: the library must be built."""
import ctypes
import os

import torch

from . import _native  # noqa: F401  (torch's HIP runtime first)

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_PKG, "libtransoar_convgemm.so")
ABI_VERSION = 3
CL3D = torch.channels_last_3d


def _load():
    if not os.path.exists(_LIB_PATH):
        raise _native.NativeLibraryError("%s is not built (python transoar_amd/_build.py)" % _LIB_PATH)
    lib = ctypes.CDLL(_LIB_PATH)
    i, p, u = ctypes.c_int, ctypes.c_void_p, ctypes.c_uint
    lib.transoar_conv3d_igemm.restype = i
    lib.transoar_conv3d_igemm.argtypes = [p] * 5 + [i] * 17 + [u] * 3 + [i, i, p]
    lib.transoar_conv3d_finish.restype = i
    lib.transoar_conv3d_finish.argtypes = [p, p, p, ctypes.c_long, i, i, p]
    lib.transoar_conv3d_wgrad.restype = i
    lib.transoar_conv3d_wgrad.argtypes = [p, p, p, p] + [i] * 10 + [u] * 3 + [i, i, p]
    lib.transoar_linear_wgrad_bias.restype = i
    lib.transoar_linear_wgrad_bias.argtypes = [p] * 6 + [i] * 4 + [p]
    lib.transoar_conv3d_wgrad_ring.restype = i
    lib.transoar_conv3d_wgrad_ring.argtypes = [p, p, p, p] + [i] * 11 + [p]
    lib.transoar_conv3d_wgrad_part_floats.restype = ctypes.c_long
    lib.transoar_conv3d_wgrad_part_floats.argtypes = [i, i, i, i]
    lib.transoar_conv3d_dgrad_s2_halo.restype = i
    lib.transoar_conv3d_dgrad_s2_halo.argtypes = [p, p, p] + [i] * 9 + [p]
    lib.transoar_conv3d_pack.restype = i
    lib.transoar_conv3d_pack.argtypes = [p, p, p, i, i, p]
    lib.transoar_conv3d_pack_many.restype = i
    lib.transoar_conv3d_pack_many.argtypes = [p, i, ctypes.c_long, p]
    lib.transoar_convgemm_abi_version.restype = i
    if lib.transoar_convgemm_abi_version() != ABI_VERSION:
        raise _native.NativeLibraryError("%s: ABI mismatch, rebuild" % _LIB_PATH)
    return lib


lib = _load()


def _taps(entries):
    """[(delta, t), ...] -> packed per-axis tap list (include/transoar_convgemm.h)."""
    v = len(entries)
    for e, (delta, t) in enumerate(entries):
        v |= ((delta + 1) << (2 + 4 * e)) | (t << (4 + 4 * e))
    return v


TAPS_FWD = _taps([(-1, 0), (0, 1), (1, 2)])        # source = s*m + t - 1
TAPS_DGRAD1 = _taps([(1, 0), (0, 1), (-1, 2)])     # source = m + 1 - t
TAPS_PARITY = (_taps([(0, 1)]), _taps([(1, 0), (0, 2)]))     # dx voxel 2m + p: p = 0 sees tap 1 of dy m; p = 1 taps 0, 2 of dy m+1, m
TAPS_ONE = _taps([(0, 1)])

# launches with fewer output tiles than this split their K steps (per-split partial maps + a finish pass)
SPLIT_BELOW_TILES = 384
DGRAD_S2_HALO = os.environ.get("TRANSOAR_DGRAD_S2_HALO", "1") != "0"
WGRAD_BLOCKS = 512          # workgroups of a weight-gradient launch: one resident set (2 per CU); 2x for very long or very wide problems


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (what, rc))


def _stream():
    return torch.cuda.current_stream().cuda_stream


def pack_fwd(weight):
    """(Cout, Cin, 3,3,3) -> (27, Cout, Cin) bf16."""
    co, ci = weight.shape[:2]
    return weight.permute(2, 3, 4, 0, 1).reshape(27, co, ci).to(torch.bfloat16).contiguous()


def pack_dgrad(weight):
    """(Cout, Cin, 3,3,3) -> (27, Cin, Cout) bf16: the data gradient contracts over Cout."""
    co, ci = weight.shape[:2]
    return weight.permute(2, 3, 4, 1, 0).reshape(27, ci, co).to(torch.bfloat16).contiguous()


def pack_both(weight):
    """fp32 contiguous (Cout, Cin, 3,3,3) -> (wk (27, Cout, Cin), wkt (27, Cin, Cout)) bf16.  TRANSOAR_CONV_PACK_HIP=1: in one
    kernel (transoar_conv3d_pack) -- measured 0.5 ms per step SLOWER than torch's two permute-and-cast copies (its 32 x 32
    tiles are 2 to 576 blocks per layer: the small layers do not fill the chip), so it is off by default."""
    co, ci = weight.shape[:2]
    if weight.dtype != torch.float32 or not weight.is_contiguous() or not os.environ.get("TRANSOAR_CONV_PACK_HIP"):
        return pack_fwd(weight), pack_dgrad(weight)
    wk = torch.empty((27, co, ci), dtype=torch.bfloat16, device=weight.device)
    wkt = torch.empty((27, ci, co), dtype=torch.bfloat16, device=weight.device)
    with torch.cuda.device(weight.device):
        _check(lib.transoar_conv3d_pack(weight.data_ptr(), wk.data_ptr(), wkt.data_ptr(), co, ci, _stream()), "transoar_conv3d_pack")
    return wk, wkt


class PackPlan:
    """Both filter packs of MANY convolution layers in one launch (transoar_conv3d_pack_many), into persistent buffers.
    modules: objects with a contiguous fp32 `.weight` (Cout, Cin, 3,3,3); after run() each module carries
    `_packs = (wk, wkt)` and `_packs_version = weight._version`, which conv3d.Conv3dK3 uses as long as the weight has not
    been written since (an optimizer step bumps the version; a captured graph replays the pack kernel with the step)."""

    def __init__(self, modules):
        self.modules = list(modules)
        assert self.modules
        dev = self.modules[0].weight.device
        self.buffers, rows, tile = [], [], 0
        for m in self.modules:
            w = m.weight
            assert w.dtype == torch.float32 and w.is_contiguous() and tuple(w.shape[2:]) == (3, 3, 3) and w.device == dev
            co, ci = w.shape[:2]
            wk = torch.empty((27, co, ci), dtype=torch.bfloat16, device=dev)
            wkt = torch.empty((27, ci, co), dtype=torch.bfloat16, device=dev)
            self.buffers.append((wk, wkt))
            rows.append([w.data_ptr(), wk.data_ptr(), wkt.data_ptr(), co | (ci << 32), tile])
            tile += ((co + 31) // 32) * ((ci + 31) // 32)
        self.total_tiles = tile
        self.ptrs = [r[0] for r in rows]
        self.table = torch.tensor(rows, dtype=torch.int64, device=dev)

    def valid(self):
        return all(m.weight.data_ptr() == p for m, p in zip(self.modules, self.ptrs))

    def run(self):
        with torch.cuda.device(self.table.device):
            _check(lib.transoar_conv3d_pack_many(self.table.data_ptr(), len(self.modules), self.total_tiles, _stream()),
                   "transoar_conv3d_pack_many")
        for m, packs in zip(self.modules, self.buffers):
            m._packs, m._packs_version = packs, m.weight._version


def _split_for(tiles, k_steps):
    if tiles >= SPLIT_BELOW_TILES:
        return 1
    return max(1, min((SPLIT_BELOW_TILES + tiles - 1) // tiles, k_steps // 4, 16))


def _tiles(rows, cout):
    bn = 64 if cout <= 64 else 128
    return ((rows + 127) // 128) * ((cout + bn - 1) // bn)


def _igemm(x, wk, bias, out, src_dims, cin, cout, m_dims, src_stride, out_dims, out_stride, parity, taps, n, split, classes=0):
    """One launch of transoar_conv3d_igemm (+ the summing / cast pass when split > 1); `out` is the whole output map."""
    b = bias.data_ptr() if bias is not None else None
    with torch.cuda.device(x.device):
        if split > 1:
            rows_out = n * out_dims[0] * out_dims[1] * out_dims[2]
            part = torch.empty((split, rows_out, cout), dtype=torch.float32, device=x.device)      # written completely
            _check(lib.transoar_conv3d_igemm(x.data_ptr(), wk.data_ptr(), None, None, part.data_ptr(), n, *src_dims, cin, cout,
                                             *m_dims, src_stride, *out_dims, out_stride, *parity, *taps, split, classes, _stream()),
                   "transoar_conv3d_igemm")
            _check(lib.transoar_conv3d_finish(part.data_ptr(), b, out.data_ptr(), rows_out, cout, split, _stream()),
                   "transoar_conv3d_finish")
        else:
            _check(lib.transoar_conv3d_igemm(x.data_ptr(), wk.data_ptr(), b, out.data_ptr(), None, n, *src_dims, cin, cout,
                                             *m_dims, src_stride, *out_dims, out_stride, *parity, *taps, 1, classes, _stream()),
                   "transoar_conv3d_igemm")


MAX_ROWS = 1 << 21          # rows (voxels of the row space) per launch: the kernels decompose a row index with float reciprocals


def _batch_chunks(n, rows_per_sample, what):
    """Batch ranges [(first, count), ...] whose row count stays below MAX_ROWS (one range when the whole batch fits).
    A channels-last (N, C, D, H, W) tensor sliced along N is still a dense channels-last tensor, so a range is just
    another launch on a view.  AMOS at its reference batch (2 x 128 x 128 x 64 = 2^21 rows at stage 1) needs this."""
    if n * rows_per_sample < MAX_ROWS:
        return [(0, n)]
    per = (MAX_ROWS - 1) // rows_per_sample
    if per < 1:
        raise RuntimeError("%s: one sample has %d rows, the implicit-GEMM kernels address < %d per launch"
                           % (what, rows_per_sample, MAX_ROWS))
    return [(i, min(per, n - i)) for i in range(0, n, per)]


def conv_forward(x, wk, bias, stride, split=None):
    """x (N, Cin, D, H, W) NDHWC bf16, wk (27, Cout, Cin) bf16, bias fp32 or None -> (N, Cout, Do, Ho, Wo) NDHWC bf16."""
    n, ci, d, h, w = x.shape
    co = wk.shape[1]
    od, oh, ow = (d - 1) // stride + 1, (h - 1) // stride + 1, (w - 1) // stride + 1
    chunks = _batch_chunks(n, od * oh * ow, "conv_forward")
    if len(chunks) > 1:
        y = torch.empty((n, co, od, oh, ow), dtype=torch.bfloat16, device=x.device, memory_format=CL3D)
        for i, c in chunks:
            y[i:i + c] = conv_forward(x[i:i + c], wk, bias, stride, split)
        return y
    y = torch.empty((n, co, od, oh, ow), dtype=torch.bfloat16, device=x.device, memory_format=CL3D)
    if split is None:
        split = _split_for(_tiles(n * od * oh * ow, co), (27 * ci + 63) // 64)
    _igemm(x, wk, bias, y, (d, h, w), ci, co, (od, oh, ow), stride, (od, oh, ow), 1, (0, 0, 0), (TAPS_FWD,) * 3, n, split)
    return y


def conv_dgrad(gy, wkt, stride, in_dims, split=None):
    """gy (N, Cout, Do, Ho, Wo) NDHWC bf16, wkt (27, Cin, Cout) bf16 -> dx (N, Cin, D, H, W) NDHWC bf16."""
    n, co, od, oh, ow = gy.shape
    ci = wkt.shape[1]
    d, h, w = in_dims
    gx = torch.empty((n, ci, d, h, w), dtype=torch.bfloat16, device=gy.device, memory_format=CL3D)
    halo = split is None and stride == 2 and DGRAD_S2_HALO and ci <= 32 and co in (16, 32, 48)
    chunks = [(0, n)] if halo else _batch_chunks(n, d * h * w, "conv_dgrad")
    if len(chunks) > 1:
        for i, c in chunks:
            gx[i:i + c] = conv_dgrad(gy[i:i + c], wkt, stride, in_dims, split)
        return gx
    if stride == 1:
        if split is None:
            split = _split_for(_tiles(n * d * h * w, ci), (27 * co + 63) // 64)
        _igemm(gy, wkt, None, gx, (od, oh, ow), co, ci, (d, h, w), 1, (d, h, w), 1, (0, 0, 0), (TAPS_DGRAD1,) * 3, n, split)
        return gx
    if split is None and DGRAD_S2_HALO and ci <= 32 and co in (16, 32, 48):
        # few input channels, full-resolution dx: one workgroup computes all eight parity classes of a dx tile (HBM bound)
        with torch.cuda.device(gy.device):
            _check(lib.transoar_conv3d_dgrad_s2_halo(gy.data_ptr(), wkt.data_ptr(), gx.data_ptr(), n, od, oh, ow, d, h, w, ci, co, _stream()),
                   "transoar_conv3d_dgrad_s2_halo")
        return gx
    # stride 2: the eight parity classes (pd, ph, pw) of the dx voxels in one launch: 1, 2, 4 or 8 taps each
    if split is None:
        split = _split_for(_tiles(n * d * h * w, ci), (8 * co + 63) // 64)      # K steps of the eight-tap class
    _igemm(gy, wkt, None, gx, (od, oh, ow), co, ci, (1, 1, 1), 1, (d, h, w), 2, (0, 0, 0), (TAPS_ONE,) * 3, n, split, classes=1)
    return gx


def _wgrad(x2, gy2, geom, taps, taps_out, out_shape):
    n, sd, sh, sw, ci, co, md, mh, mw, stride = geom
    nt = 1
    for t in taps:
        nt *= t & 3
    bt = 64 if (ci <= 64 and co <= 64) else 128
    tiles = nt * ((co + bt - 1) // bt) * ((ci + bt - 1) // bt)
    rows = n * md * mh * mw
    budget = WGRAD_BLOCKS * (2 if (rows >= (1 << 20) or tiles >= 64) else 1)      # tools/tune_wgrad.py
    chunks = max(1, min(budget // tiles, rows // 1024 if rows >= 1024 else 1))      # >= 16 K steps of 64 rows per block
    part = torch.empty(lib.transoar_conv3d_wgrad_part_floats(ci, co, chunks, taps_out), dtype=torch.float32, device=x2.device)
    dw = torch.empty(out_shape, dtype=torch.float32, device=x2.device)
    with torch.cuda.device(x2.device):
        _check(lib.transoar_conv3d_wgrad(gy2.data_ptr(), x2.data_ptr(), part.data_ptr(), dw.data_ptr(), n, sd, sh, sw, ci, co, md, mh, mw,
                                         stride, *taps, chunks, taps_out, _stream()), "transoar_conv3d_wgrad")
    return dw


WGRAD_RING = os.environ.get("TRANSOAR_WGRAD_RING", "1") != "0"
WGRAD_RING_CHUNKS = 85           # workgroups per filter plane of a ring launch: 3 x 85 = one 512-thread workgroup per CU


def wgrad_ring_supported(ci, co, ow, rows):
    """The LDS-ring weight gradient (conv_wgrad_ring.hpp): up to 64 channels on either side and more than 32 on at least
    one, W-rows of dy in 64-voxel units, enough voxels to keep one persistent workgroup per CU busy."""
    return (WGRAD_RING and ci <= 64 and co <= 64 and max(ci, co) > 32 and ci % 8 == 0 and co % 8 == 0 and ow % 64 == 0
            and rows >= (1 << 18))


def conv_wgrad_ring(x, gy, stride, chunks=None):
    n, ci, d, h, w = x.shape
    co, od, oh, ow = gy.shape[1:]
    chunks = chunks or WGRAD_RING_CHUNKS
    part = torch.empty(lib.transoar_conv3d_wgrad_part_floats(ci, co, chunks, 27), dtype=torch.float32, device=x.device)
    dw = torch.empty((co, ci, 3, 3, 3), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib.transoar_conv3d_wgrad_ring(gy.data_ptr(), x.data_ptr(), part.data_ptr(), dw.data_ptr(), n, d, h, w, ci, co, od, oh, ow,
                                              stride, chunks, _stream()), "transoar_conv3d_wgrad_ring")
    return dw


def conv_wgrad(x, gy, stride):
    """x (N, Cin, D, H, W), gy (N, Cout, Do, Ho, Wo) NDHWC bf16 -> dW (Cout, Cin, 3, 3, 3) fp32."""
    n, ci, d, h, w = x.shape
    co, od, oh, ow = gy.shape[1:]
    if wgrad_ring_supported(ci, co, ow, n * od * oh * ow):
        return conv_wgrad_ring(x, gy, stride)
    chunks = _batch_chunks(n, od * oh * ow, "conv_wgrad")
    if len(chunks) > 1:          # the filter gradient is a sum over samples
        dw = None
        for i, c in chunks:
            part = conv_wgrad(x[i:i + c], gy[i:i + c], stride)
            dw = part if dw is None else dw.add_(part)
        return dw
    return _wgrad(x, gy, (n, d, h, w, ci, co, od, oh, ow, stride), (TAPS_FWD,) * 3, 27, (co, ci, 3, 3, 3))


def linear_wgrad(x, gy):
    """x (T, K), gy (T, N) bf16 contiguous -> dW (N, K) fp32 = gy^T x (the one-tap case)."""
    t, k = x.shape
    nn_ = gy.shape[1]
    return _wgrad(x, gy, (1, 1, 1, t, k, nn_, 1, 1, t, 1), (TAPS_ONE,) * 3, 1, (nn_, k))


def linear_wgrad_bias_usable(k, n):
    """Does the last K tile have a padding column for the ones that sum gy (transoar_linear_wgrad_bias)?"""
    return k % (64 if (k <= 64 and n <= 64) else 128) != 0


def linear_wgrad_bias(x, gy):
    """x (T, K), gy (T, N) bf16 contiguous -> (dW (N, K), db (N,)) fp32 = (gy^T x, gy.sum(0)) in one pass over gy."""
    t, k = x.shape
    nn_ = gy.shape[1]
    bt = 64 if (k <= 64 and nn_ <= 64) else 128
    tiles = ((nn_ + bt - 1) // bt) * ((k + bt - 1) // bt)
    budget = WGRAD_BLOCKS * (2 if (t >= (1 << 20) or tiles >= 64) else 1)
    chunks = max(1, min(budget // tiles, t // 1024 if t >= 1024 else 1))
    part = torch.empty(lib.transoar_conv3d_wgrad_part_floats(k, nn_, chunks, 1), dtype=torch.float32, device=x.device)
    bias_part = torch.empty(chunks * nn_, dtype=torch.float32, device=x.device)
    dw = torch.empty((nn_, k), dtype=torch.float32, device=x.device)
    db = torch.empty(nn_, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib.transoar_linear_wgrad_bias(gy.data_ptr(), x.data_ptr(), part.data_ptr(), dw.data_ptr(), bias_part.data_ptr(), db.data_ptr(),
                                              t, k, nn_, chunks, _stream()), "transoar_linear_wgrad_bias")
    return dw, db


def supported(cin, cout, rows):
    return cin % 8 == 0 and cout % 8 == 0 and cin >= 8 and rows < (1 << 21)
