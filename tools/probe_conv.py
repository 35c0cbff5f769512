"""Probe: what does PyTorch/MIOpen do with the backbone's conv3d shapes on this box?"""
import sys, time, torch, torch.nn.functional as F
dev = "cuda"
shapes = [  # (Cin, Cout, D, H, W, stride)
    (1, 24, 160, 160, 256, 1), (24, 24, 160, 160, 256, 1), (24, 48, 160, 160, 256, 2), (48, 48, 80, 80, 128, 1),
    (48, 96, 80, 80, 128, 2), (96, 96, 40, 40, 64, 1), (96, 192, 40, 40, 64, 2), (192, 192, 20, 20, 32, 1),
    (96, 384, 40, 40, 64, 1),
]
def bench(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
for dt, cl in ((torch.bfloat16, False), (torch.bfloat16, True), (torch.float16, True), (torch.float32, False), (torch.float32, True)):
    for (ci, co, D, H, W, s) in shapes:
        x = torch.randn(2, ci, D, H, W, device=dev, dtype=dt, requires_grad=True)
        w = torch.randn(co, ci, 3, 3, 3, device=dev, dtype=dt, requires_grad=True)
        if cl:
            x = x.detach().contiguous(memory_format=torch.channels_last_3d).requires_grad_()
            w = w.detach().contiguous(memory_format=torch.channels_last_3d).requires_grad_()
        try:
            f = bench(lambda: F.conv3d(x, w, stride=s, padding=1))
            y = F.conv3d(x, w, stride=s, padding=1)
            g = torch.randn_like(y)
            b = bench(lambda: torch.autograd.grad(y, (x, w), g, retain_graph=True))
            flop = 2 * 27 * ci * co * y.shape[2] * y.shape[3] * y.shape[4] * 2
            print(f"{str(dt)[6:]:9s} cl={cl!s:5s} {ci:4d}->{co:4d} {D}x{H}x{W} s{s}: fwd {f:8.2f} ms ({flop/f/1e9:7.1f} TF)  bwd {b:8.2f} ms ({2*flop/b/1e9:7.1f} TF)", flush=True)
        except Exception as e:
            print(dt, cl, ci, co, "ERR", str(e)[:80], flush=True)
        del x, w
