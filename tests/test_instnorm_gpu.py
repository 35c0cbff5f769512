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
