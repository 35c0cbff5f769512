"""Fused masked cross-attention of the Focused Decoder (csrc/attn.hip through the C ABI of
include/transoar_attn.h) against the explicit masked dense attention in fp32 -- the arithmetic of
necks/focused_decoder.py:238-262 on the per-organ gathered tokens: scores, -inf on the padded keys, softmax, P v."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(b, o, r, n_keys, lengths, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    c = 384
    qf = (torch.randn(b, o, r, c, generator=g) * scale * c ** -0.5).to(torch.bfloat16).cuda()
    v = torch.randn(b, o, n_keys, c, generator=g).to(torch.bfloat16).cuda()
    k = (v.float().cpu() + torch.randn(o, n_keys, c, generator=g).cuda().cpu() * 0.5).to(torch.bfloat16).cuda()
    pad = torch.ones(o, n_keys, dtype=torch.bool)
    for i, n in enumerate(lengths):
        pad[i, :n] = False
    return qf, k, v, pad.cuda()


def _dense(qf, k, v, pad, dctx=None):
    """fp32 (fp64 accumulation is not needed at these sizes) reference on the bf16-rounded operands."""
    q32, k32, v32 = (t.float().requires_grad_(dctx is not None) for t in (qf, k, v))
    s = q32 @ k32.transpose(-1, -2)
    s = s.masked_fill(pad[None, :, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    ctx = p @ v32
    if dctx is None:
        return ctx
    ctx.backward(dctx.float())
    return ctx.detach(), q32.grad, k32.grad + v32.grad        # the key gradient goes to the values (k = v + constant)


def _rel(a, b):
    return float((a.float() - b.float()).abs().max()) / max(float(b.float().abs().max()), 1e-30)


CASES = [
    # (B, O, R, L, per-organ key counts, n_split)
    (1, 2, 40, 64, (64, 33), 1),                  # one partial row block, a partly masked tile
    (2, 3, 216, 203, (203, 150, 1), 1),           # flagship rows; L not a multiple of 32; an organ with a single key
    (2, 3, 216, 203, (203, 150, 1), 3),           # the same through the key splits + combine
    (1, 2, 130, 700, (700, 385), 2),              # two row blocks, the second with 2 live rows; splits of unequal length
    (1, 1, 216, 96, (96,), 5),                    # more splits than tiles: empty splits
]


@pytest.mark.parametrize("case", CASES)
def test_fused_roi_attention_matches_dense_masked_attention(case):
    from transoar_amd import roi_attn
    b, o, r, n_keys, lengths, split = case
    qf, k, v, pad = _inputs(b, o, r, n_keys, lengths, seed=r + n_keys)
    qf.requires_grad_(True)
    v.requires_grad_(True)
    out = roi_attn.roi_attention(qf, k, v, pad, split)
    dctx = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).cuda()
    out.backward(dctx)
    torch.cuda.synchronize()
    ref, dq_ref, dtok_ref = _dense(qf.detach(), k, v.detach(), pad, dctx)
    # bf16 outputs of fp32 arithmetic; P and dS enter the matrix cores rounded to bf16 (2^-9 each)
    assert _rel(out, ref) <= 2.0 ** -7, _rel(out, ref)
    assert _rel(qf.grad, dq_ref) <= 2.0 ** -6, _rel(qf.grad, dq_ref)
    assert _rel(v.grad, dtok_ref) <= 2.0 ** -6, _rel(v.grad, dtok_ref)
    # padded keys receive exactly zero
    assert float(v.grad[:, pad].abs().max()) == 0.0 if bool(pad.any()) else True


def test_fused_roi_attention_large_score_range_and_flagship_size():
    """The decoder's real size (2 x 20 organs, 216 folded rows, 5520 keys) with scores spread over +-40: the running
    maximum moves many times (accumulator rescaling) and most keys underflow."""
    from transoar_amd import roi_attn
    lengths = [5520 - 137 * i for i in range(20)]
    qf, k, v, pad = _inputs(2, 20, 216, 5520, lengths, seed=11, scale=12.0)
    qf.requires_grad_(True)
    v.requires_grad_(True)
    out = roi_attn.roi_attention(qf, k, v, pad)
    dctx = torch.randn(out.shape, generator=torch.Generator().manual_seed(6)).to(torch.bfloat16).cuda()
    out.backward(dctx)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(v.grad).all()) and bool(torch.isfinite(qf.grad).all())
    sel = [0, 7, 19]                           # three organs of each batch element against the dense reference
    idx = torch.tensor(sel).cuda()
    ref, dq_ref, dtok_ref = _dense(qf.detach()[:, idx], k[:, idx], v.detach()[:, idx], pad[idx], dctx[:, idx])
    assert _rel(out[:, idx], ref) <= 2.0 ** -7
    assert _rel(qf.grad[:, idx], dq_ref) <= 2.0 ** -6
    assert _rel(v.grad[:, idx], dtok_ref) <= 2.0 ** -6


def test_folded_cross_attention_takes_the_fused_kernel_and_matches_the_torch_chain():
    """focused_decoder._roi_attention_folded with the kernel against its round-3 torch chain (_FoldedCore)."""
    from transoar_amd import focused_decoder as fd
    from transoar_amd import roi_attn
    qf, k, v, pad = _inputs(2, 4, 216, 300, (300, 257, 64, 200), seed=3)
    a = qf.clone().requires_grad_(True)
    va = v.clone().requires_grad_(True)
    b_ = qf.clone().requires_grad_(True)
    vb = v.clone().requires_grad_(True)
    assert roi_attn.usable(a, k, va)
    y1 = roi_attn.roi_attention(a, k, va, pad)
    y2 = fd._FoldedCore.apply(b_, k, vb, pad)
    g = torch.randn(y1.shape, generator=torch.Generator().manual_seed(9)).to(torch.bfloat16).cuda()
    y1.backward(g)
    y2.backward(g)
    assert _rel(y1, y2) <= 2.0 ** -6
    assert _rel(a.grad, b_.grad) <= 2.0 ** -5 and _rel(va.grad, vb.grad) <= 2.0 ** -5
