#!/usr/bin/env python
"""Time weight-gradient GEMM variants for the refine block's Linear layers
(T = 234 000 tokens): dW = dY^T X.  hipBLASLt's TN kernel vs chunked bmm."""
import torch, sys
T = 234000
dev = "cuda"
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for N, K in [(384, 384), (1024, 384), (384, 1024), (288, 384), (96, 384)]:
    dy = torch.randn(T, N, device=dev, dtype=torch.bfloat16)
    x = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
    ref = (dy.float().t() @ x.float())
    base = timeit(lambda: dy.t() @ x)
    line = "N=%4d K=%4d  mm %.3f ms (%.0f TF/s)" % (N, K, base, 2e-9 * T * N * K / base)
    for B in (8, 16, 40, 80, 125):
        a = dy.view(B, T // B, N).transpose(1, 2)
        b = x.view(B, T // B, K)
        try:
            f = lambda: torch.bmm(a, b, out_dtype=torch.float32).sum(0)
            f()
        except Exception as ex:
            f = lambda: torch.bmm(a, b).float().sum(0)
        t = timeit(f)
        err = ((f() - ref).abs().max() / ref.abs().max()).item()
        line += " | B=%d %.3f (e %.1e)" % (B, t, err)
    # swapped: (X^T dY)^T
    t = timeit(lambda: (x.t() @ dy).t())
    line += " | swapped %.3f" % t
    print(line, flush=True)
