"""GPU: the implicit-GEMM 3x3x3 convolution kernels against PyTorch's fp32
convolution on the same bf16-rounded inputs.  Tolerance: outputs are rounded
to bf16 (2^-8 relative to the tensor's max); weight gradients are fp32 sums of
bf16 products (1e-3 relative)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-300)


CASES = [  # N, Cin, Cout, D, H, W, stride, bias
    (2, 24, 24, 6, 10, 16, 1, False), (1, 24, 48, 8, 8, 16, 2, False), (2, 48, 96, 4, 6, 16, 2, False),
    (1, 96, 96, 5, 5, 8, 1, False), (1, 96, 384, 4, 4, 8, 1, True), (2, 8, 40, 3, 7, 24, 1, True),
    (1, 1, 24, 6, 10, 16, 1, False), (1, 192, 64, 2, 2, 8, 1, True),
    (2, 24, 24, 3, 4, 64, 1, False), (1, 8, 32, 2, 3, 128, 1, False), (1, 32, 16, 3, 2, 64, 1, False),   # LDS-transposed wgrad
    (1, 48, 48, 3, 3, 64, 1, False), (1, 40, 64, 2, 2, 64, 1, True),    # ... in 32-channel blocks
    (2, 1, 32, 3, 4, 32, 1, False), (1, 1, 8, 2, 3, 48, 1, False), (3, 1, 24, 5, 3, 16, 1, False),   # Cin=1: MFMA weight gradient
    # >= 2^16 voxels, stride 1, <= 24 -> <= 32 channels: the LDS halo-tile kernel, partial tiles on every axis
    (1, 24, 24, 18, 22, 200, 1, False), (1, 8, 32, 20, 30, 120, 1, True), (2, 16, 24, 17, 33, 136, 1, False),
    (1, 1, 24, 18, 30, 128, 1, False),
    # Cin = 1, W % 64 == 0: the coalesced weight gradient (dy through the transposing LDS read, x rows staged per kw)
    (2, 1, 24, 5, 7, 64, 1, False), (1, 1, 32, 4, 6, 256, 1, False), (2, 1, 8, 3, 5, 192, 1, False), (1, 1, 16, 6, 4, 256, 1, True),
]


@pytest.mark.parametrize("case", CASES)
def test_conv3d_k3_forward_backward(case):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transoar_amd.conv3d import Conv3dK3
    Conv3dK3.min_voxels = 0                 # small shapes on purpose: always take the HIP path
    n, ci, co, d, h, w, s, bias = case
    torch.manual_seed(ci * 1000 + co)
    conv = Conv3dK3(ci, co, 3, stride=s, padding=1, bias=bias).cuda()
    x = torch.randn(n, ci, d, h, w, device="cuda").to(torch.bfloat16).requires_grad_(ci != 1)
    wb = conv.weight.detach().to(torch.bfloat16).float()
    y = conv(x)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last_3d)
    xr = x.detach().float().requires_grad_(ci != 1)
    wr = wb.clone().requires_grad_()
    br = conv.bias.detach().float().clone().requires_grad_() if bias else None
    yr = F.conv3d(xr, wr, br, stride=s, padding=1)
    assert tuple(y.shape) == tuple(yr.shape)
    assert relerr(y, yr) <= 2.0 ** -7
    g = torch.randn_like(yr).to(torch.bfloat16)
    y.backward(g)
    yr.backward(g.float())
    assert relerr(conv.weight.grad, wr.grad) <= 2e-3
    if bias:
        assert relerr(conv.bias.grad, br.grad) <= 2e-3
    if ci != 1:
        assert relerr(x.grad, xr.grad) <= 2.0 ** -7


def test_layout_kernels_roundtrip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transoar_amd.conv3d import to_ncdhw, to_ndhwc
    x = torch.randn(2, 24, 3, 5, 8, device="cuda").to(torch.bfloat16)
    cl = to_ndhwc(x)
    assert cl.is_contiguous(memory_format=torch.channels_last_3d) and torch.equal(cl, x)
    back = to_ncdhw(cl)
    assert back.is_contiguous() and torch.equal(back, x)


def _gpu_relerr(a, b):
    return float((a.float() - b.float()).abs().max()) / max(float(b.float().abs().max()), 1e-30)


# The flagship step's first layers at their real size (batch 2 of 160x160x256, config.py _BACKBONE): stem 1->24,
# 24->24 at full resolution, the first strided stage 24->48.  Class defaults (min_voxels, which gradients are
# hand-written) stay as the training step uses them.
# ... and one layer per deeper stage + the FPN output convolutions at their flagship shapes (round-2 VERDICT item 3):
# these run on the LDS-tiled implicit GEMM of csrc/conv_gemm.hip (strided data gradient = the parity-class launch).
@pytest.mark.parametrize("case", [(2, 1, 24, 160, 160, 256, 1), (2, 24, 24, 160, 160, 256, 1),
                                  (2, 24, 48, 160, 160, 256, 2), (2, 48, 48, 80, 80, 128, 1), (2, 48, 96, 80, 80, 128, 2),
                                  (2, 96, 96, 40, 40, 64, 1), (2, 96, 192, 40, 40, 64, 2), (2, 192, 384, 20, 20, 32, 2),
                                  (2, 384, 384, 10, 10, 16, 1), (2, 384, 768, 10, 10, 16, 2), (2, 768, 768, 5, 5, 8, 1),
                                  (2, 96, 384, 40, 40, 64, 1), (2, 384, 384, 5, 5, 8, 1),
                                  # AMOS geometry at its reference batch (round-3 ADVICE): 2 x 128 x 128 x 64 = 2^21 output
                                  # rows, one more than an implicit-GEMM launch addresses -> batch ranges (conv_gemm._batch_chunks)
                                  (2, 24, 48, 256, 256, 128, 2), (2, 48, 48, 128, 128, 64, 1)])
def test_conv3d_k3_flagship_layer_shapes(case):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transoar_amd.conv3d import Conv3dK3
    Conv3dK3.min_voxels = 0
    n, ci, co, d, h, w, s = case
    torch.manual_seed(ci + co)
    conv = Conv3dK3(ci, co, 3, stride=s, padding=1, bias=False).cuda()
    x = torch.randn(n, ci, d, h, w, device="cuda").to(torch.bfloat16)
    if ci != 1:
        x = x.contiguous(memory_format=torch.channels_last_3d)
    x.requires_grad_(ci != 1)
    y = conv(x)
    assert y.dtype == torch.bfloat16
    xr = x.detach().float().requires_grad_(ci != 1)
    wr = conv.weight.detach().to(torch.bfloat16).float().requires_grad_()
    yr = F.conv3d(xr, wr, None, stride=s, padding=1)
    assert _gpu_relerr(y, yr) <= 2.0 ** -7
    g = torch.randn(yr.shape, device="cuda").to(torch.bfloat16)
    y.backward(g)
    yr.backward(g.float())
    # 13e6-term fp32 sums of bf16 products against fp32 sums of fp32 products, different summation orders
    assert _gpu_relerr(conv.weight.grad, wr.grad) <= 5e-3
    if ci != 1:
        assert _gpu_relerr(x.grad, xr.grad) <= 2.0 ** -7
