"""GPU: the LDS-tiled implicit-GEMM convolution kernels (csrc/conv_gemm.hip) against PyTorch's fp32 convolution on the
same bf16-rounded inputs: forward (stride 1 and 2, bias, split-K), data gradient (stride 1: mirrored taps; stride 2:
eight parity-class launches, with and without the shared fp32 accumulator), weight gradient, and the token-projection
weight gradient as the one-tap case.  Tolerances: bf16 outputs 2^-7 of the tensor's max; fp32 weight gradients 2e-3."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-300)


CASES = [  # N, Cin, Cout, D, H, W, stride, bias  -- odd sizes, partial tiles on rows and channels, K steps straddling taps
    (2, 48, 48, 6, 10, 16, 1, False), (1, 24, 48, 8, 8, 16, 2, False), (2, 48, 96, 4, 6, 16, 2, False),
    (1, 96, 96, 5, 5, 8, 1, False), (1, 96, 384, 4, 4, 8, 1, True), (2, 8, 40, 3, 7, 24, 1, True),
    (1, 192, 136, 3, 5, 7, 1, True), (1, 200, 64, 5, 7, 9, 2, False), (1, 384, 768, 2, 4, 4, 2, False),
    (1, 768, 768, 2, 2, 4, 1, False), (2, 96, 192, 6, 6, 8, 2, False), (1, 56, 72, 7, 9, 11, 2, True),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("split", [None, 1, 3])
def test_conv_gemm_forward_dgrad_wgrad(case, split):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transoar_amd import conv_gemm as G
    n, ci, co, d, h, w, s, bias = case
    torch.manual_seed(ci * 1000 + co)
    x = torch.randn(n, ci, d, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    wt = (torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.1).to(torch.bfloat16).float()
    b = torch.randn(co, device="cuda") if bias else None
    xr, wr = x.float().requires_grad_(), wt.clone().requires_grad_()
    yr = F.conv3d(xr, wr, b, stride=s, padding=1)
    if True:
        y = G.conv_forward(x, G.pack_fwd(wt), b, s, split=split)
        assert y.is_contiguous(memory_format=torch.channels_last_3d) and tuple(y.shape) == tuple(yr.shape)
        assert relerr(y, yr) <= 2.0 ** -7
        g = torch.randn_like(yr).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
        yr.backward(g.float())
        gx = G.conv_dgrad(g, G.pack_dgrad(wt), s, (d, h, w), split=split)
        assert relerr(gx, xr.grad) <= 2.0 ** -7
    gw = G.conv_wgrad(x, g, s)
    assert relerr(gw, wr.grad) <= 2e-3


@pytest.mark.parametrize("case", [(1, 24, 48, 8, 8, 16), (2, 24, 48, 7, 9, 15), (1, 8, 16, 5, 6, 18), (1, 32, 32, 9, 8, 17),
                                  (2, 16, 48, 12, 10, 20), (1, 24, 48, 20, 24, 48),
                                  (1, 24, 48, 260, 8, 16)])          # D > 255: the piece table's 'no piece' marker must not alias a row
def test_stride2_dgrad_halo_kernel(case):
    """transoar_conv3d_dgrad_s2_halo (few input channels: all eight parity classes of a dx tile in one workgroup) against
    torch's fp32 data gradient and against the general parity-class launch; odd sizes: D = 2 OD - 1, partial tiles."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transoar_amd import conv_gemm as G
    n, ci, co, d, h, w = case
    torch.manual_seed(ci + co + d)
    xr = torch.zeros(n, ci, d, h, w, device="cuda", requires_grad=True)
    wt = (torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.1).to(torch.bfloat16).float()
    yr = F.conv3d(xr, wt, None, stride=2, padding=1)
    g = torch.randn_like(yr).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    yr.backward(g.float())
    wkt = G.pack_dgrad(wt)
    assert G.DGRAD_S2_HALO
    gx = G.conv_dgrad(g, wkt, 2, (d, h, w))
    general = G.conv_dgrad(g, wkt, 2, (d, h, w), split=1)
    assert gx.is_contiguous(memory_format=torch.channels_last_3d)
    assert relerr(gx, xr.grad) <= 2.0 ** -7
    assert relerr(gx, general) <= 2.0 ** -7


@pytest.mark.parametrize("case", [(1, 24, 48, 12, 20, 256, 2), (2, 48, 48, 5, 9, 128, 1), (1, 48, 24, 6, 18, 64, 1), (1, 56, 40, 7, 11, 127, 2),
                                  (1, 64, 64, 4, 17, 64, 1), (1, 8, 40, 9, 40, 255, 2)])
def test_weight_gradient_ring_kernel(case):
    """transoar_conv3d_wgrad_ring (one filter plane per workgroup, x rows in an LDS ring, 8 waves = channel-tile pairs x a
    K split) against torch's fp32 weight gradient and the general voxel-major kernel; odd source sizes, 2 and 4 tile
    pairs, stride 1 and 2, few and many workgroups per plane."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transoar_amd import conv_gemm as G
    n, ci, co, d, h, w, s = case
    torch.manual_seed(ci * 7 + co)
    x = torch.randn(n, ci, d, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    wr = torch.zeros(co, ci, 3, 3, 3, device="cuda", requires_grad=True)
    yr = F.conv3d(x.float(), wr, None, stride=s, padding=1)
    assert yr.shape[-1] % 64 == 0
    g = torch.randn_like(yr).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    yr.backward(g.float())
    got = G.conv_wgrad_ring(x, g, s, chunks=3 if ci == 56 else None)
    assert relerr(got, wr.grad) <= 2e-3
    general = G._wgrad(x, g, (n, d, h, w, ci, co) + tuple(yr.shape[2:]) + (s,), (G.TAPS_FWD,) * 3, 27, (co, ci, 3, 3, 3))
    assert relerr(got, general) <= 2e-3


@pytest.mark.parametrize("t,k,nn_", [(1000, 384, 384), (4097, 384, 1024), (333, 1024, 384), (5000, 48, 64)])
def test_linear_wgrad_is_the_one_tap_case(t, k, nn_):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transoar_amd import conv_gemm as G
    torch.manual_seed(t)
    x = torch.randn(t, k, device="cuda").to(torch.bfloat16)
    gy = torch.randn(t, nn_, device="cuda").to(torch.bfloat16)
    want = gy.float().t() @ x.float()
    got = G.linear_wgrad(x, gy)
    assert relerr(got, want) <= 2e-3


@pytest.mark.parametrize("co,ci", [(48, 24), (96, 96), (200, 56), (768, 384)])
def test_filter_pack_kernel(co, ci):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transoar_amd import conv_gemm as G
    import os
    w = torch.randn(co, ci, 3, 3, 3, device="cuda")
    os.environ["TRANSOAR_CONV_PACK_HIP"] = "1"
    try:
        wk, wkt = G.pack_both(w)
    finally:
        del os.environ["TRANSOAR_CONV_PACK_HIP"]
    assert torch.equal(wk, G.pack_fwd(w)) and torch.equal(wkt, G.pack_dgrad(w))


def test_pack_plan_packs_many_layers_in_one_launch():
    """conv_gemm.PackPlan: one launch for the filter packs of several layers == the per-layer torch packs, and
    Conv3dK3 uses them exactly as long as the weight version matches."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transoar_amd import conv_gemm as G
    from transoar_amd.conv3d import Conv3dK3
    torch.manual_seed(3)
    mods = [Conv3dK3(ci, co, 3, stride=s, padding=1, bias=False).cuda() for ci, co, s in [(24, 48, 2), (48, 48, 1), (96, 200, 2), (40, 56, 1)]]
    plan = G.PackPlan(mods)
    plan.run()
    for m in mods:
        wk, wkt = m._packs
        assert m._packs_version == m.weight._version
        assert torch.equal(wk, G.pack_fwd(m.weight.detach())) and torch.equal(wkt, G.pack_dgrad(m.weight.detach()))
    m = mods[1]
    x = torch.randn(1, 48, 4, 6, 8, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y_plan = m(x)
        with torch.no_grad():
            m.weight.mul_(2.0)                    # version bump: the packs are stale and must not be used
        assert m._packs_version != m.weight._version
        y_new = m(x)
    assert torch.allclose(y_new.float(), 2 * y_plan.float(), rtol=2e-2, atol=1e-3)
    assert plan.valid()
