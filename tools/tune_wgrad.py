import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from transoar_amd import conv_gemm as G
def time_ms(fn, iters=20):
    for _ in range(3): fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))[iters // 2]
for m, k, n in ((234000, 384, 384), (234000, 384, 1024), (234000, 1024, 384)):
    x = torch.randn(m, k, device="cuda").bfloat16(); gy = torch.randn(m, n, device="cuda").bfloat16()
    out = {}
    for wb in (512, 1024, 2048, 4096, 8192):
        G.WGRAD_BLOCKS = wb
        out[wb] = round(time_ms(lambda: G.linear_wgrad(x, gy)), 4)
    print(m, k, n, out, flush=True)
# conv layers
for name, ci, co, d, h, w, s in (("s2c2", 96, 96, 40, 40, 64, 1), ("outP2", 96, 384, 40, 40, 64, 1), ("s1c1", 24, 48, 160, 160, 256, 2), ("s3c1", 96, 192, 40, 40, 64, 2)):
    x = torch.randn(2, ci, d, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    gy = torch.randn(2, co, d // s, h // s, w // s, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    out = {}
    for wb in (512, 1024, 2048, 4096, 8192):
        G.WGRAD_BLOCKS = wb
        out[wb] = round(time_ms(lambda: G.conv_wgrad(x, gy, s), 10), 4)
    print(name, out, flush=True)
