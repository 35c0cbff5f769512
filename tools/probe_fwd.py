#!/usr/bin/env python
"""Forward gather only, flagship shape, bf16: time per call for one location distribution (for counter passes and
kernel variants selected through the environment).

    python tools/probe_fwd.py [--dist model|init|uniform|wide] [--iters 10] [--fused]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _inputs  # noqa: E402
from transoar_amd import MSDA  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dist", default="model")
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    jitter = 0.0 if args.dist == "init" else 0.3
    value, shapes, lsi, loc, attn = _inputs.model_like_inputs(0, 2, _inputs.VISCERAL_LEVELS, device="cuda", jitter=jitter)
    if args.dist == "uniform":
        loc = torch.rand_like(loc)
    if args.dist == "wide":
        loc = (loc - 0.5) * 1.3 + 0.5 + 0.05 * torch.randn_like(loc)
    v = value.to(torch.bfloat16)
    for _ in range(3):
        MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, attn, 64)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.iters + 1)]
    ev[0].record()
    for i in range(args.iters):
        MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, attn, 64)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.iters))
    print(json.dumps({"dist": args.dist, "probe": os.environ.get("TRANSOAR_PCM_PROBE", "0"), "ms_median": round(ts[len(ts) // 2], 4)}), flush=True)


if __name__ == "__main__":
    main()
