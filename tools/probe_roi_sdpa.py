#!/usr/bin/env python
"""RoI cross-attention core: explicit bmm/softmax (current) vs F.scaled_dot_product_attention
on strided head views (dev probe)."""
import torch, torch.nn.functional as F
B, O, h, qpo, L, hd = 2, 20, 8, 27, 5500, 48
C = h * hd
dev = "cuda"
torch.manual_seed(0)
q = torch.randn(B, O * qpo, C, device=dev, dtype=torch.bfloat16, requires_grad=True)
k = torch.randn(B, O * L, C, device=dev, dtype=torch.bfloat16, requires_grad=True)
v = torch.randn(B, O * L, C, device=dev, dtype=torch.bfloat16, requires_grad=True)
pad = torch.rand(O, L, device=dev) > 0.8
pad[:, 0] = False
def explicit():
    kk = k.view(B, O, L, h, hd).permute(0, 1, 3, 2, 4)
    vv = v.view(B, O, L, h, hd).permute(0, 1, 3, 2, 4)
    qq = q.view(B, O, qpo, h, hd).permute(0, 1, 3, 2, 4)
    attn = qq @ kk.transpose(-2, -1)
    attn = attn.masked_fill(pad[None, :, None, None, :], float("-inf")).softmax(dim=-1)
    return (attn @ vv).permute(0, 1, 3, 2, 4).reshape(B, O * qpo, C)
bias = torch.zeros(O, L, device=dev, dtype=torch.bfloat16).masked_fill(pad, float("-inf"))
def sdpa(mask_kind="bias"):
    kk = k.view(B, O, L, h, hd).permute(0, 1, 3, 2, 4).reshape(B * O, h, L, hd)
    vv = v.view(B, O, L, h, hd).permute(0, 1, 3, 2, 4).reshape(B * O, h, L, hd)
    qq = q.view(B, O, qpo, h, hd).permute(0, 1, 3, 2, 4).reshape(B * O, h, qpo, hd)
    if mask_kind == "bias":
        m = bias[None, :, None, None, :].expand(B, O, 1, 1, L).reshape(B * O, 1, 1, L)
    else:
        m = (~pad)[None, :, None, None, :].expand(B, O, 1, 1, L).reshape(B * O, 1, 1, L)
    x = F.scaled_dot_product_attention(qq, kk, vv, attn_mask=m, scale=1.0)
    return x.view(B, O, h, qpo, hd).permute(0, 1, 3, 2, 4).reshape(B, O * qpo, C)
def timeit(f, n=10):
    go = torch.randn(B, O * qpo, C, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        f().backward(go)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f().backward(go)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
qs = q.detach() * hd ** -0.5
ref = explicit()
print("explicit fwd+bwd ms", timeit(explicit))
for kind in ("bias", "bool"):
    try:
        out = sdpa(kind)
        print(kind, "sdpa fwd+bwd ms", timeit(lambda: sdpa(kind)), "max diff", (out - ref).abs().max().item(), ref.abs().max().item())
    except Exception as ex:
        print(kind, "failed:", repr(ex)[:300])
