"""GPU: fused InstanceNorm3d(affine)+ReLU kernels against PyTorch fp32 on the
same bf16-rounded inputs (outputs rounded to bf16: 2^-7 relative to max)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-300)


@pytest.mark.parametrize("shape", [(2, 24, 6, 10, 16), (1, 48, 5, 7, 8), (2, 96, 4, 4, 8), (1, 768, 2, 3, 8),
                                   (3, 192, 1, 1, 8)])
@pytest.mark.parametrize("relu", [True, False])
def test_instnorm_relu(shape, relu):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transoar_amd.instnorm import instance_norm_relu, supported
    torch.manual_seed(shape[1])
    x = (torch.randn(shape, device="cuda") * 2 + 0.5).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last_3d).requires_grad_()
    assert supported(x, shape[1])
    gamma = (1 + 0.2 * torch.randn(shape[1], device="cuda")).requires_grad_()
    beta = (0.3 * torch.randn(shape[1], device="cuda")).requires_grad_()
    y = instance_norm_relu(x, gamma, beta, 1e-5, relu)
    xr = x.detach().float().requires_grad_()
    gr, br = gamma.detach().clone().requires_grad_(), beta.detach().clone().requires_grad_()
    yr = F.instance_norm(xr, weight=gr, bias=br, eps=1e-5)
    if relu:
        yr = F.relu(yr)
    assert relerr(y, yr) <= 2.0 ** -7
    g = torch.randn_like(yr).to(torch.bfloat16)
    y.backward(g)
    yr.backward(g.float())
    assert relerr(x.grad, xr.grad) <= 2e-2      # bf16 dx; the relu mask can flip on rounding ties
    assert relerr(gamma.grad, gr.grad) <= 1e-2
    assert relerr(beta.grad, br.grad) <= 1e-2


def test_instnorm_relu_flagship_stem_shape():
    """The largest InstanceNorm of the step: batch 2 x 24 channels x 160x160x256 (6.5e6 voxels per instance)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transoar_amd.instnorm import instance_norm_relu, supported
    torch.manual_seed(24)
    shape = (2, 24, 160, 160, 256)
    x = (torch.randn(shape, device="cuda") * 2 + 0.5).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last_3d).requires_grad_()
    assert supported(x, 24)
    gamma = (1 + 0.2 * torch.randn(24, device="cuda")).requires_grad_()
    beta = (0.3 * torch.randn(24, device="cuda")).requires_grad_()
    y = instance_norm_relu(x, gamma, beta, 1e-5, True)
    xr = x.detach().float().requires_grad_()
    gr, br = gamma.detach().clone().requires_grad_(), beta.detach().clone().requires_grad_()
    yr = F.relu(F.instance_norm(xr, weight=gr, bias=br, eps=1e-5))
    err = lambda a, b: float((a.float() - b.float()).abs().max()) / float(b.float().abs().max())
    assert err(y, yr) <= 2.0 ** -7
    g = torch.randn(shape, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    y.backward(g)
    yr.backward(g.float())
    assert err(x.grad, xr.grad) <= 2e-2
    # per-channel sums over 13e6 voxels: fp32 accumulation order differs
    assert err(gamma.grad, gr.grad) <= 1e-2
    assert err(beta.grad, br.grad) <= 1e-2


@pytest.mark.parametrize("cin,shape", [(1, (2, 32, 48, 64)), (24, (2, 30, 34, 72)), (24, (1, 41, 44, 40)), (8, (1, 32, 48, 64))])
def test_statistics_from_the_conv_epilogue(cin, shape):
    """conv3d_k3_lds<.., STATS>: the per-channel sum / sum of squares of the convolution's bf16 output, taken from the
    accumulators, against sums over the stored output; and InstanceNorm + ReLU fed with them against the kernel that makes
    its own pass (same apply pass: only the statistics' summation order differs), forward and backward."""
    from transoar_amd import conv3d, instnorm
    n, d, h, w = shape
    g = torch.Generator(device="cuda").manual_seed(cin + d)
    conv = conv3d.Conv3dK3(cin, 24, 3, padding=1, bias=False).cuda()
    x = torch.randn(n, cin, d, h, w, device="cuda", generator=g).to(torch.bfloat16)
    if cin > 1:
        x = x.contiguous(memory_format=torch.channels_last_3d)
    if cin > 1:
        x.requires_grad_()                    # (the one-channel stem's input is the volume: no data gradient exists for it)
    gamma = (torch.rand(24, device="cuda", generator=g) + 0.5).requires_grad_()
    beta = torch.randn(24, device="cuda", generator=g).requires_grad_()
    y, part = conv.forward_with_stats(x)
    assert part is not None and part.shape[1:] == (2, 32)
    y_plain = conv(x)
    assert torch.equal(y, y_plain)
    yf = y.detach().float()
    rows = part.shape[0] // n
    got = part.view(n, rows, 2, 32).double().sum(1)
    want_sum = yf.double().sum((2, 3, 4))
    want_sq = (yf.double() ** 2).sum((2, 3, 4))
    scale = (yf.double().abs()).sum((2, 3, 4))
    assert ((got[:, 0, :24] - want_sum).abs() <= 2e-6 * scale).all()
    assert ((got[:, 1, :24] - want_sq).abs() <= 2e-6 * want_sq).all()
    assert float(got[:, :, 24:].abs().max()) == 0.0
    out = instnorm.instance_norm_relu(y, gamma, beta, 1e-5, True, part)
    ref = instnorm.instance_norm_relu(y_plain, gamma, beta, 1e-5, True)
    assert float((out.float() - ref.float()).abs().max()) <= 2.0 ** -7 * float(ref.float().abs().max())
    assert float((out != ref).float().mean()) < 1e-3           # same apply pass: an entry moves only where a statistic's last bits flip a rounding
    go = torch.randn(out.shape, device="cuda", generator=g).to(torch.bfloat16)
    wrt = ((x,) if cin > 1 else ()) + (gamma, beta, conv.weight)
    grads = torch.autograd.grad(out, wrt, go, retain_graph=True)
    grads_ref = torch.autograd.grad(ref, wrt, go)
    for a, b in zip(grads, grads_ref):
        assert float((a.float() - b.float()).abs().max()) <= 1e-2 * float(b.float().abs().max())
