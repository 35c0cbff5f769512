#!/usr/bin/env python
"""The Swin MLP's two GEMMs with the GELU in their epilogues (gemm.linear_gelu / linear_gelu_grad) against the same GEMMs
followed by torch's gelu / gelu_backward, at the stage-0 shape of BASELINE config #4 (2 x 819 200 tokens, 48 -> 192 -> 48).
One JSON line per variant."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transoar_amd import gemm  # noqa: E402


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
    for t, c, hid in ((2 * 819200, 48, 192), (2 * 102400, 96, 384), (2 * 12800, 192, 768)):
        g = torch.Generator().manual_seed(0)
        x = torch.randn(t, c, generator=g).to(torch.bfloat16).cuda()
        w1 = (torch.randn(hid, c, generator=g) * c ** -0.5).to(torch.bfloat16).cuda()
        w2t = (torch.randn(hid, c, generator=g) * hid ** -0.5).to(torch.bfloat16).cuda()      # fc2's weight transposed: (hid, c)
        b1 = torch.randn(hid, generator=g).cuda()
        gy = torch.randn(t, c, generator=g).to(torch.bfloat16).cuda()
        h, a = gemm.linear_gelu(x, w1, b1)
        rec = {"op": "gelu_mlp", "tokens": t, "C": c, "hidden": hid}
        rec["fc1_plain_ms"] = round(time_ms(lambda: gemm.linear_nt(x, w1, b1)), 4)
        rec["gelu_ms"] = round(time_ms(lambda: torch.nn.functional.gelu(h)), 4)
        rec["fc1_gelu_fused_ms"] = round(time_ms(lambda: gemm.linear_gelu(x, w1, b1)), 4)
        ga = gemm.linear_nt(gy, w2t)
        rec["fc2_dgrad_plain_ms"] = round(time_ms(lambda: gemm.linear_nt(gy, w2t)), 4)
        rec["gelu_backward_ms"] = round(time_ms(lambda: torch.ops.aten.gelu_backward(ga, h)), 4)
        rec["fc2_dgrad_gelu_fused_ms"] = round(time_ms(lambda: gemm.linear_gelu_grad(gy, w2t, h)), 4)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
