"""token_linear (chunked weight-gradient GEMM) against autocast nn.Linear."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tokens,n_in,n_out", [(2 * 20000, 384, 384), (33001, 384, 1024), (40000, 1024, 96)])
@pytest.mark.parametrize("in_dtype", [torch.float32, torch.bfloat16])
def test_matches_autocast_linear(tokens, n_in, n_out, in_dtype):
    from transoar_amd.token_linear import token_linear, MIN_TOKENS
    assert tokens >= MIN_TOKENS
    torch.manual_seed(0)
    lin = torch.nn.Linear(n_in, n_out).cuda()
    x = torch.randn(1, tokens, n_in, device="cuda").to(in_dtype)
    gy = torch.randn(1, tokens, n_out, device="cuda", dtype=torch.bfloat16)
    res = []
    for fn in (lambda t: lin(t), lambda t: token_linear(t, lin.weight, lin.bias)):
        xi = x.clone().requires_grad_(True)
        lin.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = fn(xi)
        y.backward(gy)
        res.append((y, xi.grad, lin.weight.grad.clone(), lin.bias.grad.clone()))
    (y0, gx0, gw0, gb0), (y1, gx1, gw1, gb1) = res
    assert y1.dtype == torch.bfloat16 and gx1.dtype == in_dtype and gw1.dtype == torch.float32
    from transoar_amd import token_linear as tl
    if tl.LAST_PATH == "blas":
        assert torch.equal(y0, y1)
        assert torch.equal(gx0, gx1)
    else:
        # hand-written GEMM (K = N = 384): the bias is added in fp32 before the single rounding to bf16 and the
        # K loop runs in another order -> a result may differ from hipBLASLt's by one bf16 ulp; both are
        # measured against fp64 of the same bf16 operands
        xb, wb = x[0].to(torch.bfloat16).double(), lin.weight.detach().to(torch.bfloat16).double()
        ty = xb @ wb.t() + lin.bias.detach().double()
        e0, e1 = (y0[0].double() - ty).abs().max(), (y1[0].double() - ty).abs().max()
        assert e1 <= max(e0, 2.0 ** -8 * ty.abs().max())
        tgx = gy[0].double() @ wb
        g0, g1 = (gx0[0].double() - tgx).abs().max(), (gx1[0].double() - tgx).abs().max()
        assert g1 <= max(g0, 2.0 ** -8 * tgx.abs().max())
    # fp64 truth for the weight gradient: both paths must be equally close to it
    truth = gy[0].double().t() @ x[0].to(torch.bfloat16).double()
    scale = truth.abs().max()
    assert (gw1.double() - truth).abs().max() <= max(2.0 * (gw0.double() - truth).abs().max(), 1e-5 * scale)
    tb = gy[0].double().sum(0)
    assert (gb1.double() - tb).abs().max() <= max(2.0 * (gb0.double() - tb).abs().max(), 1e-5 * tb.abs().max())


def test_small_inputs_take_the_stock_path():
    from transoar_amd.token_linear import token_linear
    lin = torch.nn.Linear(8, 8).cuda()
    x = torch.randn(2, 10, 8, device="cuda", requires_grad=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = token_linear(x, lin.weight, lin.bias)
    assert "TokenLinear" not in type(y.grad_fn).__name__
