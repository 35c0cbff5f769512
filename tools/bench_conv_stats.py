#!/usr/bin/env python
"""conv3d_k3_lds with and without the InstanceNorm statistics in its epilogue (24 -> 24 and the stem at 160x160x256)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transoar_amd import conv3d as C


def t_ms(fn, n=10):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))[n // 2]


for ci in (24, 1):
    x = torch.randn(2, ci, 160, 160, 256, device="cuda").to(torch.bfloat16)
    if ci > 1:
        x = x.contiguous(memory_format=torch.channels_last_3d)
    wt = torch.randn(24, ci, 3, 3, 3, device="cuda") * 0.05
    wk = C._pack_taps(torch.nn.functional.pad(wt, (0, 0, 0, 0, 0, 0, 0, 7)) if ci == 1 else wt)
    plain = t_ms(lambda: C.conv3d_k3_forward_c1(x, wk, None) if ci == 1 else C.conv3d_k3_forward(x, wk, None, 1))
    stats = t_ms(lambda: C.conv3d_k3_forward_stats(x, wk, None))
    print(json.dumps({"cin": ci, "plain_ms": round(plain, 4), "with_stats_ms": round(stats, 4)}))
