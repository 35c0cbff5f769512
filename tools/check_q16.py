#!/usr/bin/env python
"""Round 6's forward gather (msda3d_q16.hpp, 16 queries per wave) (flag 128) against round 3's point-column kernel (the default) and the
per-corner brick kernel (flag 16) on the flagship pyramid: differences and times per location distribution.

    python tools/check_q16.py [--geometry visceral|amos] [--n 2] [--iters 20] [--dists model,init,uniform,wide] [--time-only]

TRANSOAR_MSDA3D_Q16_UPW (units per wave) and TRANSOAR_MSDA3D_Q16_PROBE (1: no parameter stream, 2: no geometry either --
measurement only, results are not the operator's) are read by the library once per process.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _inputs  # noqa: E402
from transoar_amd import MSDA  # noqa: E402

Q16, NO_MMA = 128, 16


def timed(fn, iters):
    for _ in range(3):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--geometry", default="visceral")
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dists", default="model,init,uniform,wide")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--time-only", action="store_true")
    args = ap.parse_args()
    levels = _inputs.VISCERAL_LEVELS if args.geometry == "visceral" else _inputs.AMOS_LEVELS
    vdt = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    tag = {"upw": os.environ.get("TRANSOAR_MSDA3D_Q16_UPW", "4"), "probe": os.environ.get("TRANSOAR_MSDA3D_Q16_PROBE", "0")}
    bad = 0
    for dist in args.dists.split(","):
        jitter = 0.0 if dist == "init" else 0.3
        value, shapes, lsi, loc, attn = _inputs.model_like_inputs(0, args.n, levels, device="cuda", jitter=jitter)
        if dist == "uniform":
            loc = torch.rand_like(loc)
        if dist == "wide":
            loc = (loc - 0.5) * 1.3 + 0.5 + 0.05 * torch.randn_like(loc)
        v = value.to(vdt)
        kernels = (("q16", Q16),) if args.time_only else (("q16", Q16), ("pcm", 0), ("brick", NO_MMA))
        out = {}
        for name, fl in kernels:
            MSDA.flags = fl
            out[name] = MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, attn, 64).float()
            ms = timed(lambda: MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, attn, 64), args.iters)
            print(json.dumps({"dist": dist, "kernel": name, "ms": round(ms, 4), **(tag if name == "q16" else {})}), flush=True)
        MSDA.flags = 0
        if args.time_only:
            continue
        scale = out["brick"].abs().max().item()
        for a in ("q16", "pcm"):
            d = (out[a] - out["brick"]).abs().max().item()
            print(json.dumps({"dist": dist, "diff": a + " vs brick", "max_abs": d, "rel_to_max": d / scale}), flush=True)
        d = (out["q16"] - out["pcm"]).abs().max().item()
        print(json.dumps({"dist": dist, "diff": "q16 vs pcm", "max_abs": d, "rel_to_max": d / scale,
                          "nan": bool(torch.isnan(out["q16"]).any())}), flush=True)
        if not (out["q16"] - out["brick"]).abs().max().item() <= 2.0 ** -7 * scale:
            bad += 1
            e = (out["q16"] - out["pcm"]).abs()
            N, S = e.shape[0], e.shape[1]
            rows = (e.amax(-1) > 2.0 ** -7 * scale).nonzero()
            print(json.dumps({"dist": dist, "bad_rows": int(rows.shape[0]), "of": N * S, "first": rows[:12].tolist()}), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
