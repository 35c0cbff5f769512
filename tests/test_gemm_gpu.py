"""GPU: the hand-written bf16/f16 MFMA token GEMM (csrc/gemm.hip) against a plain PyTorch fp32 matmul of the
same 16-bit-rounded operands.  fp32 accumulation both sides -> only the output rounding (and summation order)
differs: 2^-8 (bf16) / 2^-11 (f16) of the largest |value|; fp32 output 1e-5."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x, w, bias, relu):
    y = x.float() @ w.float().t()
    if bias is not None:
        y = y + bias
    return y.relu() if relu else y


# (234000 .. 100001 rows: the streaming kernels of csrc/gemm_stream.hip -- K = 384 / N = 384, >= 16 384 dense rows; odd heights;
# the last four: at most 64 output columns -- the 128 x 64 tile of the Swin blocks' narrow products, bf16)
@pytest.mark.parametrize("m,k,n", [(1000, 384, 384), (4099, 1024, 384), (300, 384, 1024), (777, 64, 100), (128, 8, 4),
                                    (234000, 384, 384), (234000, 384, 1024), (234000, 1024, 384), (100001, 384, 576),
                                    (100001, 768, 384), (50001, 192, 48), (30000, 48, 48), (9000, 144, 64), (40007, 96, 60)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_gemm_nt_matches_fp32_matmul(m, k, n, dt, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from transoar_amd import gemm
    monkeypatch.setattr(gemm, "STREAM_N384", True)          # exercise the N = 384 streaming kernel too (not the default route)
    g = torch.Generator(device="cuda").manual_seed(m + k + n)
    x = torch.randn(m, k, device="cuda", generator=g).to(dt)
    w = (torch.randn(n, k, device="cuda", generator=g) / k ** 0.5).to(dt)
    b = torch.randn(n, device="cuda", generator=g)
    tol = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    for bias, relu in ((None, False), (b, False), (b, True)):
        want = _ref(x, w, bias, relu)
        got = gemm.linear_nt(x, w, bias, relu)
        assert got.dtype == dt and got.shape == (m, n)
        if m >= 16384 and dt == torch.bfloat16 and (k == 384) != (n == 384):
            assert gemm.stream_kind(x, w) == ("k384" if k == 384 else "n384")
        assert float((got.float() - want).abs().max()) <= tol * float(want.abs().max()) + 1e-6
    got32 = gemm.linear_nt(x, w, b, False, out_dtype=torch.float32)
    assert float((got32 - _ref(x, w, b, False)).abs().max()) <= 1e-5 * float(_ref(x, w, b, False).abs().max()) + 1e-6


def test_gemm_nt_strided_operands_and_asymmetry():
    """Row-strided views (lda > K) and an asymmetric B: a row/column swap in the C write cannot pass."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from transoar_amd import gemm
    big = torch.arange(520 * 96, device="cuda", dtype=torch.float32).view(520, 96).remainder(7).sub(3).to(torch.bfloat16)
    x = big[:, 16:80]                                      # (520, 64), lda 96
    w = torch.zeros(36, 64, device="cuda", dtype=torch.bfloat16)
    w[torch.arange(36), torch.arange(36)] = torch.arange(1, 37, device="cuda").to(torch.bfloat16)     # scaled selector
    got = gemm.linear_nt(x, w)
    want = x.float()[:, :36] * torch.arange(1, 37, device="cuda")
    assert torch.equal(got.float(), want.to(torch.bfloat16).float())


def test_token_linear_runs_on_the_hand_written_gemm():
    """token_linear (the refine block's projections under bf16 autocast): forward and input gradient come from
    csrc/gemm.hip, the weight gradient from the chunked reduction; all three against fp32 autograd."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from transoar_amd import token_linear as tl
    tl.USE_HIP_GEMM = True
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(2, 40000, 384, device="cuda", generator=g).to(torch.bfloat16).requires_grad_()
    lin = torch.nn.Linear(384, 1024).cuda()
    gy = torch.randn(2, 40000, 1024, device="cuda", generator=g).to(torch.bfloat16)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = tl.token_linear(x, lin.weight, lin.bias)
    assert tl.LAST_PATH == "hip-gemm"
    y.backward(gy)
    xr = x.detach().float().requires_grad_()
    wr = lin.weight.detach().to(torch.bfloat16).float().requires_grad_()
    yr = torch.nn.functional.linear(xr, wr, lin.bias.detach())
    yr.backward(gy.float())
    rel = lambda a, b: float((a.float() - b).abs().max() / b.abs().max())
    assert rel(y, yr) <= 2.0 ** -7
    assert rel(x.grad, xr.grad) <= 2.0 ** -7
    assert rel(lin.weight.grad, wr.grad) <= 2e-3
    assert rel(lin.bias.grad, gy.float().sum((0, 1))) <= 1e-3
    tl.USE_HIP_GEMM = None


def test_streaming_k384_square_and_fused_relu_dropout():
    """The K = 384 streaming kernel on the square projection shape (not its default there) and with bias + ReLU + seeded
    dropout in its epilogue: the mask is the one tokens.relu_dropout derives from the same seed (tokens.hashed_keep), so
    the fused linear1 equals relu_dropout(linear) up to the GEMMs' output rounding; its autograd against fp32."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from transoar_amd import gemm, token_linear as tl, tokens
    g = torch.Generator(device="cuda").manual_seed(5)
    m = 50001
    x = torch.randn(m, 384, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(1024, 384, device="cuda", generator=g) / 384 ** 0.5).to(torch.bfloat16)
    b = torch.randn(1024, device="cuda", generator=g)
    gemm.STREAM_SQUARE = True
    try:
        w2 = w[:384].contiguous()
        assert gemm.stream_kind(x, w2) == "k384"
        got = gemm.linear_nt(x, w2, b[:384])
        want = _ref(x, w2, b[:384], False)
        assert float((got.float() - want).abs().max()) <= 2.0 ** -8 * float(want.abs().max())
    finally:
        gemm.STREAM_SQUARE = False
    seed = tokens.dropout_seed(x)
    keep_prob = 0.9
    y = gemm.linear_relu_dropout(x, w, b, seed, keep_prob)
    keep = tokens.hashed_keep(seed, m * 1024, keep_prob).view(m, 1024).float()
    want = _ref(x, w, b, True) * keep / keep_prob
    assert float((y.float() - want).abs().max()) <= 2.0 ** -8 * float(want.abs().max())
    assert float(y.float()[keep == 0].abs().max()) == 0.0               # dropped elements are exactly zero
    assert abs(float(keep.mean()) - keep_prob) < 2e-3
    # autograd of the fused layer (module interface) against fp32 with the same mask
    lin = torch.nn.Linear(384, 1024).cuda()
    drop = torch.nn.Dropout(0.1).train()
    xg = x.clone().requires_grad_()
    torch.manual_seed(11)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert tl.linear_relu_dropout_usable(xg, lin.weight)
        yy = tl.linear_relu_dropout(xg, lin.weight, lin.bias, drop)
    gy = torch.randn(yy.shape, device="cuda", generator=g).to(torch.bfloat16)
    yy.backward(gy)
    torch.manual_seed(11)
    seed2 = tokens.dropout_seed(x)                                         # the seed the call above drew
    keep2 = tokens.hashed_keep(seed2, m * 1024, 0.9).view(m, 1024).float()
    xr = x.float().requires_grad_()
    wr = lin.weight.detach().to(torch.bfloat16).float().requires_grad_()
    pre = torch.nn.functional.linear(xr, wr, lin.bias.detach())
    # the gate of the reference is the kernel's own (y > 0): a pre-activation within bf16 rounding of zero may fall on
    # either side, and ONE flipped element moves a row of the weight gradient by |gy x| ~ 1 % of its largest entry
    gate = (yy.detach().float() > 0)
    assert int((gate != ((pre.detach() > 0) & (keep2 > 0))).sum()) <= 8
    yr = pre * gate.float() / 0.9
    yr.backward(gy.float())
    rel = lambda a, c: float((a.float() - c).abs().max() / c.abs().max())
    assert rel(yy, yr) <= 2.0 ** -7
    assert rel(xg.grad, xr.grad) <= 3e-2          # K = 1024 products of the bf16-rounded hidden gradient (observed 0.021)
    assert rel(lin.weight.grad, wr.grad) <= 3e-3


@pytest.mark.parametrize("t,n,k", [(20000, 384, 384), (33001, 1024, 384), (33001, 384, 1024), (16384, 512, 384), (17000, 384, 128),
                                   (70001, 128, 384)])
def test_wgrad384_matches_fp32(t, n, k):
    """dW = gy^T x over the tokens on the token-streaming kernel (csrc/gemm_stream.hip, wgrad384_kernel): both output
    orientations, both column-tile widths, token counts that are not a multiple of the 32-token stage."""
    from transoar_amd import gemm, token_linear
    g = torch.Generator(device="cuda").manual_seed(t + n)
    gy = torch.randn(t, n, device="cuda", generator=g).bfloat16()
    x = torch.randn(t, k, device="cuda", generator=g).bfloat16()
    assert gemm.wgrad384_shapes(gy, x) > 0
    dw = gemm.wgrad384(gy, x)
    if gemm.wgrad384_usable(gy, x):
        assert torch.equal(dw, token_linear.weight_grad(gy, x))
    assert dw.dtype == torch.float32 and dw.shape == (n, k)
    ref = gy.float().t() @ x.float()
    # bf16 products are exact in fp32; only the summation order differs
    assert (dw - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    # and the round-3 path (one-tap conv GEMM) agrees
    from transoar_amd import conv_gemm
    old = conv_gemm.linear_wgrad(x, gy)
    assert (dw - old).abs().max().item() <= 2e-3 * ref.abs().max().item()
    assert torch.equal(dw, gemm.wgrad384(gy, x))          # deterministic: fixed chunking, no atomics
    # the bias gradient from the same pass: column sums of gy out of the fragments the product holds in registers
    dw2, db = gemm.wgrad384(gy, x, with_bias=True)
    assert torch.equal(dw2, dw)
    want = gy.double().sum(0)
    assert (db.double() - want).abs().max().item() <= 1e-5 * gy.double().abs().sum(0).max().item()


def test_fused_ffn_node_matches_fp32_chain():
    """token_linear._FusedFFN (linear1 + ReLU + dropout + linear2 as one autograd node; the gate's gradient in the epilogue of
    linear2's data-gradient GEMM, both bias gradients out of the weight-gradient passes) against the fp32 chain that uses the
    kernel's own gate."""
    from transoar_amd import token_linear as tl, tokens
    g = torch.Generator(device="cuda").manual_seed(8)
    m = 40001
    x = torch.randn(m, 384, device="cuda", generator=g).to(torch.bfloat16)
    lin1 = torch.nn.Linear(384, 1024).cuda()
    lin2 = torch.nn.Linear(1024, 384).cuda()
    drop = torch.nn.Dropout(0.1).train()
    xg = x.clone().requires_grad_()
    torch.manual_seed(13)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert tl.fused_ffn_usable(xg, lin1.weight, lin2.weight)
        out = tl.fused_ffn(xg, lin1, lin2, drop)
        torch.manual_seed(13)
        hidden = tl.linear_relu_dropout(x, lin1.weight, lin1.bias, drop)          # same seed: the hidden tensor of the node
    gy = torch.randn(out.shape, device="cuda", generator=g).to(torch.bfloat16)
    out.backward(gy)
    gate = (hidden.detach().float() > 0).float()
    xr = x.float().requires_grad_()
    w1 = lin1.weight.detach().to(torch.bfloat16).float().requires_grad_()
    w2 = lin2.weight.detach().to(torch.bfloat16).float().requires_grad_()
    b1 = lin1.bias.detach().clone().requires_grad_()
    b2 = lin2.bias.detach().clone().requires_grad_()
    h_ref = torch.nn.functional.linear(xr, w1, b1) * gate / 0.9
    # the node's second layer reads the bf16-rounded hidden tensor
    out_ref = torch.nn.functional.linear(h_ref + (hidden.detach().float() - h_ref).detach(), w2, b2)
    out_ref.backward(gy.float())
    rel = lambda a, c: float((a.detach().float() - c.detach()).abs().max() / c.detach().abs().max())
    assert rel(hidden, h_ref) <= 2.0 ** -7
    assert rel(out, out_ref) <= 2.0 ** -7
    assert rel(xg.grad, xr.grad) <= 3e-2
    assert rel(lin1.weight.grad, w1.grad) <= 3e-3 and rel(lin2.weight.grad, w2.grad) <= 3e-3
    assert rel(lin1.bias.grad, b1.grad) <= 3e-3 and rel(lin2.bias.grad, b2.grad) <= 1e-4
