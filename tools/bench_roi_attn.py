#!/usr/bin/env python
"""Fused masked cross-attention (csrc/attn.hip) against round 3's torch chain (_FoldedCore: hipBLASLt + aten) at the
Focused Decoder's flagship size: 2 x 20 organs, 216 folded rows, 5520 keys, 384 channels.  One JSON line per variant."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transoar_amd import focused_decoder as fd  # noqa: E402
from transoar_amd import roi_attn  # noqa: E402


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
    b, o, r, n_keys, c = 2, 20, 216, 5520, 384
    g = torch.Generator().manual_seed(0)
    qf = (torch.randn(b, o, r, c, generator=g) * c ** -0.5).to(torch.bfloat16).cuda().requires_grad_(True)
    v = torch.randn(b, o, n_keys, c, generator=g).to(torch.bfloat16).cuda().requires_grad_(True)
    k = (v.detach() + 0.5).contiguous()
    pad = torch.ones(o, n_keys, dtype=torch.bool)
    for i in range(o):
        pad[i, : n_keys - 150 * i] = False
    pad = pad.cuda()
    dctx = torch.randn(b, o, r, c, generator=g).to(torch.bfloat16).cuda()
    keys = int((~pad).sum()) * b
    flops_fwd = 2 * 2 * r * c * keys
    for name, fn in (("fused", lambda: roi_attn.roi_attention(qf, k, v, pad)), ("torch_chain", lambda: fd._FoldedCore.apply(qf, k, v, pad))):
        f_ms = time_ms(lambda: fn())
        y = fn()
        def fb():
            qf.grad = v.grad = None
            fn().backward(dctx)
        fb_ms = time_ms(fb)
        print(json.dumps({"op": "roi_attention", "impl": name, "B": b, "O": o, "R": r, "L": n_keys, "C": c,
                          "fwd_ms": round(f_ms, 4), "fwd_bwd_ms": round(fb_ms, 4), "bwd_ms": round(fb_ms - f_ms, 4),
                          "fwd_TFLOPs": round(flops_fwd / f_ms / 1e9, 1),
                          "bwd_TFLOPs_5gemm": round(2.5 * flops_fwd / max(fb_ms - f_ms, 1e-6) / 1e9, 1)}), flush=True)
        del y


if __name__ == "__main__":
    main()
