import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transoar_amd import gemm
for t, n, k in ((33001, 1024, 384), (80000, 1024, 384), (33001, 384, 1024), (16384, 256, 384)):
    g = torch.Generator(device="cuda").manual_seed(t + n)
    gy = torch.randn(t, n, device="cuda", generator=g).bfloat16()
    x = torch.randn(t, k, device="cuda", generator=g).bfloat16()
    dw, db = gemm.wgrad384(gy, x, with_bias=True)
    want = gy.double().sum(0).float()
    err = (db - want).abs()
    print(t, n, k, "max err", err.max().item(), "want max", want.abs().max().item())
    print("  db  ", db[:6].tolist(), db[250:262].tolist())
    print("  want", want[:6].tolist(), want[250:262].tolist())
    # is it a partial sum? ratio / subset test
    half = gy[: t // 2].double().sum(0).float()
    print("  first-half sums", half[:6].tolist())
    bad = (err > 1e-2 * want.abs().max()).nonzero().flatten()
    print("  bad cols", bad.numel(), bad[:20].tolist())
