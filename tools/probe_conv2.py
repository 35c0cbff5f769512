import sys, time, torch, torch.nn.functional as F
dev = "cuda"
def bench(fn, n=2):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
cases = [(torch.float32, False, 24, 24), (torch.bfloat16, False, 32, 32), (torch.bfloat16, True, 32, 32),
         (torch.bfloat16, False, 16, 16), (torch.float16, False, 24, 24), (torch.bfloat16, False, 24, 32), (torch.bfloat16, False, 32, 24)]
for dt, cl, ci, co in cases:
    D, H, W = 160, 160, 256
    x = torch.randn(1, ci, D, H, W, device=dev, dtype=dt)
    w = torch.randn(co, ci, 3, 3, 3, device=dev, dtype=dt)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last_3d); w = w.contiguous(memory_format=torch.channels_last_3d)
    x.requires_grad_(); w.requires_grad_()
    f = bench(lambda: F.conv3d(x, w, padding=1))
    y = F.conv3d(x, w, padding=1); g = torch.randn_like(y)
    t0 = time.perf_counter(); torch.autograd.grad(y, x, g, retain_graph=True); torch.cuda.synchronize(); bd1 = (time.perf_counter() - t0) * 1e3
    bd = bench(lambda: torch.autograd.grad(y, x, g, retain_graph=True), 1) if bd1 < 200 else bd1
    t0 = time.perf_counter(); torch.autograd.grad(y, w, g, retain_graph=True); torch.cuda.synchronize(); bw1 = (time.perf_counter() - t0) * 1e3
    bw = bench(lambda: torch.autograd.grad(y, w, g, retain_graph=True), 1) if bw1 < 200 else bw1
    print(f"{str(dt)[6:]:9s} cl={cl!s:5s} {ci}->{co} N=1: fwd {f:8.2f}  dgrad {bd:8.2f}  wgrad {bw:8.2f} ms", flush=True)
    del x, w, y, g
