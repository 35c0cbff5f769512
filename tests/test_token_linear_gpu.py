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


def test_fpn_pointwise_convs_as_token_gemms():
    """The FPN decoder's 1x1x1 lateral and k = s = 2 transposed convolutions on the hand-written GEMM
    (backbone._conv1_as_gemm / _up2_as_gemm) against the stock convolutions: outputs and gradients within bf16
    rounding of each other."""
    import copy
    from tests import _inputs
    from transoar_amd import backbone
    torch.manual_seed(0)
    cfg = _inputs.small_backbone_config(False, levels=("P2", "P3", "P4", "P5"))
    cfg.update(start_channels=8, out_fmaps=["P2", "P3", "P4", "P5"])
    dec = backbone.Decoder(cfg).cuda()
    shapes = [(8, 32, 32, 64), (16, 16, 16, 32), (32, 8, 8, 16), (64, 4, 4, 8), (128, 2, 2, 4), (256, 1, 1, 2)]
    feats0 = {"C%d" % i: torch.randn(2, *s, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
              for i, s in enumerate(shapes)}
    res = []
    for flag in (True, False):
        backbone.Decoder.gemm_pointwise = flag
        d = copy.deepcopy(dec)
        feats = {k: v.clone().requires_grad_() for k, v in feats0.items()}
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = d(feats)
        loss = sum((o.float() * torch.linspace(-1, 1, o.numel(), device="cuda").view_as(o)).sum() for o in out.values())
        loss.backward()
        grads = {n: p.grad.clone() for n, p in d.named_parameters() if p.grad is not None}
        res.append((out, grads, {k: v.grad for k, v in feats.items() if v.grad is not None}))
    backbone.Decoder.gemm_pointwise = True
    (o1, g1, x1), (o0, g0, x0) = res
    rel = lambda a, b: float((a.float() - b.float()).abs().max()) / max(float(b.float().abs().max()), 1e-20)
    assert set(o1) == set(o0) and set(g1) == set(g0) and set(x1) == set(x0) and len(x1) >= 4
    for k in o0:
        assert rel(o1[k], o0[k]) <= 2.0 ** -6, k
    for k in g0:
        assert rel(g1[k], g0[k]) <= 2.0 ** -5, k
    for k in x0:
        assert rel(x1[k], x0[k]) <= 2.0 ** -5, k


def test_ffn_with_hidden_width_384_takes_the_unfused_path():
    """A refinement block with dim_feedforward == d_model == 384: gemm.stream_kind keeps 384 x 384 products on the tiled
    kernel, so the fused-FFN predicates must say no (round-4 advisor: they said yes on `hidden % 64 == 0` and the fused
    nodes then raised in forward and backward) and the layer's token_linear + relu_dropout branch must run."""
    from transoar_amd import gemm, tokens
    from transoar_amd.token_linear import fused_ffn_usable, linear_relu_dropout_usable, token_linear
    torch.manual_seed(0)
    lin1, lin2 = torch.nn.Linear(384, 384).cuda(), torch.nn.Linear(384, 384).cuda()
    wide = torch.nn.Linear(384, 1024).cuda()
    x = torch.randn(1, 20480, 384, device="cuda").to(torch.bfloat16).requires_grad_()
    drop = torch.nn.Dropout(0.0)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert linear_relu_dropout_usable(x, wide.weight) == bool(gemm.STREAM)          # the flagship width still fuses
        if not gemm.STREAM_SQUARE:
            assert not linear_relu_dropout_usable(x, lin1.weight)
            assert not fused_ffn_usable(x, lin1.weight, lin2.weight)
        # every answer of the predicate is one gemm.stream_kind agrees with
        for lin in (lin1, wide):
            if linear_relu_dropout_usable(x, lin.weight):
                assert gemm.stream_kind(x[0], lin.weight.to(torch.bfloat16)) == "k384"
        hidden = token_linear(x, lin1.weight, lin1.bias)
        hidden = tokens.relu_dropout(hidden, drop)
        y = token_linear(hidden, lin2.weight, lin2.bias)
    ref = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(x.float(), lin1.weight, lin1.bias)), lin2.weight, lin2.bias)
    assert (y.float() - ref).abs().max() <= 2.0 ** -5 * ref.abs().max()
    y.float().sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad.float()).all() and lin1.weight.grad is not None


@pytest.mark.parametrize("tokens,c,hidden", [(2 * 10000 + 37, 48, 192), (9000, 96, 384), (8200, 192, 776)])
def test_gelu_mlp_matches_the_unfused_chain(tokens, c, hidden):
    """fc2(gelu(fc1(x))) with the activation in the GEMM epilogues (token_linear._GeluMlp, transoar_gemm_nt_gelu) against the same
    two GEMMs with torch's gelu between them: the epilogues apply the activation to the bf16-rounded product in fp32 (Phi(x) to
    1.5e-7), so values and gradients agree to an ulp of the bf16 results.  192 / 776 hidden channels: a partial last column tile."""
    from transoar_amd.token_linear import gelu_mlp, gelu_mlp_usable, token_linear
    torch.manual_seed(1)
    fc1, fc2 = torch.nn.Linear(c, hidden).cuda(), torch.nn.Linear(hidden, c).cuda()
    x = torch.randn(1, tokens, c, device="cuda").to(torch.bfloat16)
    gy = torch.randn(1, tokens, c, device="cuda", dtype=torch.bfloat16)
    res = []
    for fused in (False, True):
        xi = x.clone().requires_grad_(True)
        fc1.zero_grad(); fc2.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if fused:
                assert gelu_mlp_usable(xi, fc1, fc2, 8192)
                y = gelu_mlp(xi, fc1, fc2)
            else:
                hid = torch.nn.functional.gelu(token_linear(xi, fc1.weight, fc1.bias, force_hip=True, min_tokens=8192))
                y = token_linear(hid, fc2.weight, fc2.bias, force_hip=True, min_tokens=8192)
        y.backward(gy)
        res.append((y.float(), xi.grad.float(), fc1.weight.grad.clone(), fc1.bias.grad.clone(), fc2.weight.grad.clone(),
                    fc2.bias.grad.clone()))
    for name, a, b in zip(("y", "gx", "gw1", "gb1", "gw2", "gb2"), res[0], res[1]):
        tol = 2.0 ** -7 if name in ("y", "gx") else 2e-3          # one bf16 ulp of the largest value / sums of ~1e4 such terms
        assert (a - b).abs().max().item() <= tol * a.abs().max().item(), name
    frac = (res[0][0] != res[1][0]).float().mean().item()
    assert frac <= 0.05, frac          # ... and nearly all outputs are bit-identical


@pytest.mark.parametrize("tokens,k,n", [(20037, 48, 144), (9001, 48, 48), (30000, 192, 48), (12345, 96, 384), (8192, 192, 768)])
def test_linear_wgrad_with_the_bias_gradient_in_its_padding_column(tokens, k, n):
    """conv_gemm.linear_wgrad_bias (transoar_linear_wgrad_bias): dW = gy^T x and db = sum_t gy from ONE pass over gy -- a column
    of ones in the x tile's padding makes the bias gradient column K of the product -- against fp64 of the same bf16 operands,
    and dW bit-identical to the kernel without the ones column."""
    from transoar_amd import conv_gemm
    g = torch.Generator().manual_seed(tokens)
    x = torch.randn(tokens, k, generator=g).to(torch.bfloat16).cuda()
    gy = (torch.randn(tokens, n, generator=g) + 0.25).to(torch.bfloat16).cuda()
    assert conv_gemm.linear_wgrad_bias_usable(k, n)
    dw, db = conv_gemm.linear_wgrad_bias(x, gy)
    assert torch.equal(dw, conv_gemm.linear_wgrad(x, gy))
    tw, tb = gy.double().t() @ x.double(), gy.double().sum(0)
    assert (dw.double() - tw).abs().max().item() <= 1e-5 * tw.abs().max().item()
    assert (db.double() - tb).abs().max().item() <= 1e-5 * tb.abs().max().item()
