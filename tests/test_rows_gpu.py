"""GPU: row gather / inverse-gather kernels vs torch indexing."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rows_gather_and_pull_sum(dtype):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transoar_amd import rows
    g = torch.Generator().manual_seed(0)
    B, S, C, K = 2, 500, 48, 1300
    x = torch.randn(B, S, C, generator=g).to(dtype).cuda()
    index = torch.randint(0, S, (K,), generator=g)
    out = rows.gather(x, index.int().cuda())
    assert torch.equal(out, x[:, index.cuda()])
    # one level cut out of a larger token matrix: read in place (batch stride = the matrix's rows), no copy
    big = torch.randn(B, S + 77, C, generator=g).to(dtype).cuda()
    level = big[:, 40:40 + S]
    assert rows.row_dense(level) and not level.is_contiguous()
    assert torch.equal(rows.gather(level, index.int().cuda()), level[:, index.cuda()])
    assert torch.equal(rows.gather(big[:, :S:2], index.int().cuda()[:10] // 2),
                       big[:, :S:2][:, (index[:10] // 2).cuda()])          # not row-dense: falls back to a copy
    # CSR inverse
    order = torch.argsort(index, stable=True)
    counts = torch.bincount(index, minlength=S)
    ptr = torch.cat((counts.new_zeros(1), counts.cumsum(0))).int().cuda()
    gy = torch.randn(B, K, C, generator=g).to(dtype).cuda()
    got = rows.pull_sum(gy, ptr, order.int().cuda(), S)
    ref = torch.zeros(B, S, C, device="cuda").index_add_(1, index.cuda(), gy.float())
    tol = 1e-6 if dtype == torch.float32 else 2 ** -7
    assert float((got.float() - ref).abs().max()) <= tol * float(ref.abs().max())


@pytest.mark.parametrize("shape", [(234000, 384), (50000, 1024), (4096, 8), (70001, 24), (9999, 2048)])
def test_colsum_matches_fp32_sum(shape):
    """Column sums of a bf16 matrix (bias gradients): fp32 accumulation, against torch's fp32 sum of the same values
    (different summation order: 1e-5 relative to the column's absolute sum)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transoar_amd import rows
    torch.manual_seed(shape[1])
    x = torch.randn(shape, device="cuda").to(torch.bfloat16)
    assert rows.colsum_usable(x)
    got = rows.colsum(x)
    ref = x.double().sum(0)
    bound = 1e-5 * x.double().abs().sum(0)
    assert got.dtype == torch.float32 and got.shape == (shape[1],)
    assert bool(((got.double() - ref).abs() <= bound + 1e-6).all())


@pytest.mark.parametrize("shape,dtype", [((1080, 384), torch.bfloat16), ((1080, 1024), torch.bfloat16),
                                         ((3240, 384), torch.bfloat16), ((7, 8), torch.bfloat16),
                                         ((1025, 2048), torch.bfloat16), ((4096, 2304), torch.float32),
                                         ((1024, 1536), torch.float32), ((300, 12), torch.float32),
                                         ((1, 4), torch.float32), ((20000, 96), torch.float32)])
def test_colsum_small_matches_fp32_sum(shape, dtype):
    """One-launch column sums of short matrices (Focused Decoder bias gradients, per-wave partials of the token kernels)."""
    from transoar_amd import rows
    torch.manual_seed(shape[0])
    x = torch.randn(shape, device="cuda").to(dtype)
    assert rows.colsum_small_usable(x)
    got = rows.colsum_small(x)
    ref = x.double().sum(0)
    bound = 1e-5 * x.double().abs().sum(0)
    assert got.dtype == torch.float32 and got.shape == (shape[1],)
    assert bool(((got.double() - ref).abs() <= bound + 1e-6).all())
    # the dispatcher: same numbers whichever kernel it picks; torch for what neither kernel takes
    any_ = rows.colsum_any(x)
    assert bool(((any_.double() - ref).abs() <= bound + 1e-6).all())
    odd = torch.randn(33, 10, device="cuda")
    assert torch.allclose(rows.colsum_any(odd), odd.sum(0))


def test_rows_gather_axpy_is_the_merge_with_its_residual():
    """out = resid + scale[b] * src[:, index] in one pass (the Swin block's window merge, csrc/rows.hip rows_gather_axpy):
    fp32 arithmetic on the bf16 operands, one rounding; a negative index reads a zero row; either of scale / resid may be absent."""
    from transoar_amd import rows
    g = torch.Generator().manual_seed(1)
    B, S, C, K = 3, 700, 48, 650
    x = torch.randn(B, S, C, generator=g).to(torch.bfloat16).cuda()
    resid = torch.randn(B, K, C, generator=g).to(torch.bfloat16).cuda()
    index = torch.randint(0, S, (K,), generator=g)
    index[::17] = -1
    scale = torch.tensor([0.0, 1.25, 1.0]).cuda()
    idx = index.int().cuda()
    picked = torch.where((index >= 0).cuda()[None, :, None], x[:, index.clamp_min(0).cuda()].float(), torch.zeros((), device="cuda"))
    want = (resid.float() + scale[:, None, None] * picked).to(torch.bfloat16)
    assert torch.equal(rows.gather_axpy(x, idx, scale, resid), want)
    assert torch.equal(rows.gather_axpy(x, idx, None, resid), (resid.float() + picked).to(torch.bfloat16))
    assert torch.equal(rows.gather_axpy(x, idx, scale, None), (scale[:, None, None] * picked).to(torch.bfloat16))
    assert torch.equal(rows.gather_axpy(x, idx), picked.to(torch.bfloat16))
