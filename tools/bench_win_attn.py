#!/usr/bin/env python
"""Swin 3-D window attention (csrc/attn.hip: win_attn_fwd / win_attn_bwd) at the four encoder stages of BASELINE config #4
(160 x 160 x 256 input, batch 2, 5 x 5 x 5 windows, head dimension 16): kernel times against the bytes the two
kernels have to move (qkv + out forward; qkv, out, dout, dqkv backward).  One JSON line per stage.
    python tools/bench_win_attn.py [--shifted] [--stage K]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transoar_amd import win_attn  # noqa: E402
from transoar_amd.swin_encoder import window_layout  # noqa: E402


def time_ms(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ts[len(ts) // 2]


def main():
    shifted = "--shifted" in sys.argv
    only = int(sys.argv[sys.argv.index("--stage") + 1]) if "--stage" in sys.argv else None
    stages = [((80, 80, 128), 48, 3), ((40, 40, 64), 96, 6), ((20, 20, 32), 192, 12), ((10, 10, 16), 384, 24)]
    g = torch.Generator().manual_seed(0)
    for k, (grid, c, heads) in enumerate(stages):
        if only is not None and k != only:
            continue
        lay = window_layout(grid, (5, 5, 5), (2, 2, 2) if shifted else (0, 0, 0), "cuda")
        b, n_win, n = 2, lay.n_windows, lay.n_per
        qkv = torch.randn(b, n_win, n, 3 * c, generator=g).to(torch.bfloat16).cuda().requires_grad_(True)
        bias = (0.02 * torch.randn(heads, n, n, generator=g)).cuda().requires_grad_(True)
        dout = torch.randn(b, n_win, n, c, generator=g).to(torch.bfloat16).cuda()
        scale = (c // heads) ** -0.5
        fn = lambda: win_attn.window_attention(qkv, bias, lay.mask_bits, heads, scale)       # noqa: E731
        f_ms = time_ms(fn)

        def fb():
            qkv.grad = bias.grad = None
            fn().backward(dout)
        fb_ms = time_ms(fb)
        tok_bytes = b * n_win * n * c * 2
        print(json.dumps({"op": "window_attention", "stage": k, "grid": grid, "C": c, "heads": heads, "windows": b * n_win,
                          "shifted": shifted, "fwd_ms": round(f_ms, 4), "bwd_ms": round(fb_ms - f_ms, 4),
                          "fwd_GBps": round(4 * tok_bytes / f_ms / 1e6, 1), "bwd_GBps": round(8 * tok_bytes / (fb_ms - f_ms) / 1e6, 1)}),
              flush=True)


if __name__ == "__main__":
    main()
