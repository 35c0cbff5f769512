"""Swin 3-D window attention kernels (csrc/attn.hip through include/transoar_attn.h) against the
explicit fp32 formulation of WindowAttention3D.forward (backbones/encoder_blocks.py:259-285): scale q k^T + relative
position bias + shifted-window mask, softmax, P v -- outputs and the gradients of qkv and of the bias."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(qkv, bias, mask, heads, scale, dout=None):
    b, nw, n, c3 = qkv.shape
    c = c3 // 3
    x = qkv.float().requires_grad_(dout is not None)
    bias32 = bias.float().detach().requires_grad_(dout is not None)
    q, k, v = x.view(b, nw, n, 3, heads, c // heads).permute(3, 0, 1, 4, 2, 5)          # (B, nW, h, n, hd)
    s = (q * scale) @ k.transpose(-1, -2) + bias32[None, None]
    if mask is not None:
        s = s + mask[None, :, None]
    out = (torch.softmax(s, dim=-1) @ v).transpose(2, 3).reshape(b, nw, n, c)
    if dout is None:
        return out
    out.backward(dout.float())
    return out.detach(), x.grad, bias32.grad


def _rel(a, b):
    return float((a.float() - b.float()).abs().max()) / max(float(b.float().abs().max()), 1e-30)


@pytest.mark.parametrize("head_dim", [16, 32])
@pytest.mark.parametrize("case", [(2, 8, 125, 3, True), (1, 3, 40, 6, False), (2, 2, 125, 24, True), (1, 5, 100, 3, True),
                                  (3, 300, 125, 3, True)])
def test_window_attention_kernels_match_the_explicit_formulation(case, head_dim):
    from transoar_amd import win_attn
    b, nw, n, heads, shifted = case
    c = heads * head_dim
    g = torch.Generator().manual_seed(n + heads)
    qkv = torch.randn(b, nw, n, 3 * c, generator=g).to(torch.bfloat16).cuda().requires_grad_(True)
    bias = (0.5 * torch.randn(heads, n, n, generator=g)).cuda().requires_grad_(True)
    mask = bits = None
    if shifted:
        label = torch.randint(0, 3, (nw, n), generator=g)
        mask = torch.zeros(nw, n, n).masked_fill_(label[:, None, :] != label[:, :, None], -100.0).cuda()
        bits = win_attn.mask_bits(mask)
    scale = head_dim ** -0.5
    assert win_attn.usable(qkv, heads)
    out = win_attn.window_attention(qkv, bias, bits, heads, scale)
    dout = torch.randn(out.shape, generator=g).to(torch.bfloat16).cuda()
    out.backward(dout)
    torch.cuda.synchronize()
    ref, dqkv_ref, dbias_ref = _reference(qkv.detach(), bias, mask, heads, scale, dout)
    assert _rel(out, ref) <= 2.0 ** -7, _rel(out, ref)
    assert _rel(qkv.grad, dqkv_ref) <= 2.0 ** -6, _rel(qkv.grad, dqkv_ref)
    # the bias gradient is a sum over batch and windows of fp32 dS values (atomics: order varies)
    assert _rel(bias.grad, dbias_ref) <= 2.0 ** -7, _rel(bias.grad, dbias_ref)
