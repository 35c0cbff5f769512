#!/usr/bin/env python
"""RoI attention with the key axis split into G groups run as extra batch entries of the
memory-efficient attention op, merged with their log-sum-exps (dev probe)."""
import torch, torch.nn.functional as F
B, O, h, qpo, L, hd = 2, 20, 8, 27, 5504, 48
C = h * hd
dev = "cuda"
torch.manual_seed(0)
q = (torch.randn(B, O * qpo, C, device=dev) * 0.3).to(torch.bfloat16).requires_grad_(True)
k = torch.randn(B, O * L, C, device=dev, dtype=torch.bfloat16, requires_grad=True)
v = torch.randn(B, O * L, C, device=dev, dtype=torch.bfloat16, requires_grad=True)
lens = torch.randint(L // 3, L, (O,), device=dev)
pad = torch.arange(L, device=dev)[None, :] >= lens[:, None]
op = torch.ops.aten._scaled_dot_product_efficient_attention
op_bwd = torch.ops.aten._scaled_dot_product_efficient_attention_backward

class SplitAttn(torch.autograd.Function):
    """groups of keys as extra batch entries in ONE op call"""
    @staticmethod
    def forward(ctx, qq, kg, vg, bias_g, G):
        # qq (N,h,Q,hd); kg,vg (N*G,h,Lg,hd) strided views; bias_g (N*G,1,1,Lg)
        N, H, Q, D = qq.shape
        Lg = kg.shape[2]
        qg = qq[:, None].expand(N, G, H, Q, D).reshape(N * G, H, Q, D)
        o, lse_g, _, _ = op(qg, kg, vg, bias_g.expand(N * G, H, Q, Lg), True, 0.0, False, scale=1.0)
        lse_g = lse_g[..., :Q].reshape(N, G, H, Q)
        lse = torch.logsumexp(lse_g, 1)                                   # (N,H,Q)
        w = torch.exp(lse_g - lse[:, None]).unsqueeze(-1)                 # (N,G,H,Q,1)
        out = (o.view(N, G, H, Q, D).float() * w).sum(1).to(qq.dtype)
        ctx.save_for_backward(qg, kg, vg, bias_g, out, lse)
        ctx.G = G
        return out
    @staticmethod
    def backward(ctx, go):
        qg, kg, vg, bias_g, out, lse = ctx.saved_tensors
        G = ctx.G
        NG, H, Q, D = qg.shape
        N = NG // G
        Lg = kg.shape[2]
        seed = torch.zeros((), dtype=torch.long, device=qg.device)
        ex = lambda t: t[:, None].expand(N, G, *t.shape[1:]).reshape(NG, *t.shape[1:])
        dq, dk, dv, _ = op_bwd(ex(go), qg, kg, vg, bias_g.expand(NG, H, Q, Lg), ex(out), ex(lse).contiguous(), seed, seed,
                               0.0, [True, True, True, False], False, scale=1.0)
        return dq.view(N, G, H, Q, D).float().sum(1).to(go.dtype), dk, dv, None, None

def views():
    kk = k.view(B, O, L, h, hd).permute(0, 1, 3, 2, 4).reshape(B * O, h, L, hd)
    vv = v.view(B, O, L, h, hd).permute(0, 1, 3, 2, 4).reshape(B * O, h, L, hd)
    qq = q.view(B, O, qpo, h, hd).permute(0, 1, 3, 2, 4).reshape(B * O, h, qpo, hd)
    return qq, kk, vv
def base():
    qq, kk, vv = views()
    keep = (~pad)[None, :, None, None, :].expand(B, O, 1, 1, L).reshape(B * O, 1, 1, L)
    return F.scaled_dot_product_attention(qq, kk, vv, attn_mask=keep, scale=1.0)
bias = torch.zeros(O, L, device=dev, dtype=torch.bfloat16).masked_fill(pad, float("-inf"))
bias = bias[None, :, None, None, :].expand(B, O, 1, 1, L).reshape(B * O, 1, 1, L).contiguous()
def split(G):
    Lg = L // G
    kg = k.view(B, O * G, Lg, h, hd).permute(0, 1, 3, 2, 4).reshape(B * O * G, h, Lg, hd)
    vg = v.view(B, O * G, Lg, h, hd).permute(0, 1, 3, 2, 4).reshape(B * O * G, h, Lg, hd)
    qq = q.view(B, O, qpo, h, hd).permute(0, 1, 3, 2, 4).reshape(B * O, h, qpo, hd)
    bg = bias.view(B * O * G, 1, 1, Lg)
    return SplitAttn.apply(qq, kg, vg, bg, G)
def timeit(f, n=10):
    go = torch.randn(B * O, h, qpo, hd, device=dev, dtype=torch.bfloat16)
    for _ in range(3): f().backward(go)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f().backward(go)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
go = torch.randn(B * O, h, qpo, hd, device=dev, dtype=torch.bfloat16)
ref = base(); ref.backward(go); gref = [t.grad.clone() for t in (q, k, v)]
print("base fwd+bwd ms %.3f" % timeit(base))
for G in (1, 2, 4, 8, 16):
    try:
        for t in (q, k, v): t.grad = None
        out = split(G); out.backward(go)
        errs = [((t.grad.float() - r.float()).abs().max() / r.float().abs().max()).item() for t, r in zip((q, k, v), gref)]
        print("G=%d fwd+bwd ms %.3f out err %.3e grads %s" % (G, timeit(lambda: split(G)),
              ((out.float() - ref.float()).abs().max() / ref.float().abs().max()).item(), ["%.2e" % e for e in errs]))
    except Exception as ex:
        print("G=%d failed: %r" % (G, ex))
