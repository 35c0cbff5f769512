#!/usr/bin/env python
"""Matrix-core forward gather (msda3d_fwd_mma) against the C oracle and the per-corner brick kernel,
plus timings at the flagship shape.  One JSON line per check.

    python tools/check_mma.py [--quick]
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import msda3d_oracle as c_oracle  # noqa: E402  (checker only)
from tests import _inputs  # noqa: E402
from transoar_amd import MSDA  # noqa: E402

NO_MMA = 16


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def fwd(v, shapes, lsi, loc, attn, flags):
    MSDA.flags = flags
    try:
        return MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, attn, 64)
    finally:
        MSDA.flags = 0


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
    return ts[len(ts) // 2], ts[0]


def small_cases():
    f = lambda t: t.float().cpu().numpy()
    for name, levels in (("3lvl_odd", [(9, 6, 11), (5, 3, 6), (2, 2, 3)]),
                         ("4lvl", [(8, 8, 16), (4, 4, 8), (2, 2, 4), (1, 1, 2)]),
                         ("1lvl_big", [(12, 20, 40)]),
                         ("amos_like", [(16, 16, 8), (8, 8, 4), (4, 4, 2)])):
        value, shapes, lsi, loc, attn = _inputs.model_like_inputs(5, 2, levels, device="cuda")
        ref_pts = _inputs.reference_points(shapes.cpu()).to("cuda")
        variants = {
            "local": loc,
            "nojitter": _inputs.model_like_inputs(5, 2, levels, device="cuda", jitter=0.0)[3],
            "uniform_oob": torch.rand_like(loc) * 1.4 - 0.2,
            "wide": (loc + (torch.rand_like(loc) - 0.5) * 0.6).contiguous(),
        }
        for vdt in (torch.bfloat16, torch.float16):
            v = value.to(vdt)
            for vn, locs in variants.items():
                a = fwd(v, shapes, lsi, locs, attn, 0)
                b = fwd(v, shapes, lsi, locs, attn, NO_MMA)
                ref = torch.from_numpy(c_oracle.forward(f(v), shapes.cpu().numpy(), lsi.cpu().numpy(), f(locs), f(attn)))
                print(json.dumps({"case": name, "variant": vn, "dtype": str(vdt), "mma_vs_oracle": relerr(a, ref),
                                  "brick_vs_oracle": relerr(b, ref), "mma_vs_brick": relerr(a, b),
                                  "nan": bool(torch.isnan(a.float()).any())}), flush=True)


def flagship(n=2, dists=("model", "model_nojitter", "uniform")):
    levels = _inputs.VISCERAL_LEVELS
    for dist in dists:
        jitter = 0.0 if dist == "model_nojitter" else 0.3
        value, shapes, lsi, loc, attn = _inputs.model_like_inputs(0, n, levels, device="cuda", jitter=jitter)
        if dist == "uniform":
            loc = torch.rand_like(loc)
        v = value.to(torch.bfloat16)
        a = fwd(v, shapes, lsi, loc, attn, 0)
        b = fwd(v, shapes, lsi, loc, attn, NO_MMA)
        err = relerr(a, b)
        # sampled queries against the C oracle (bf16-rounded value, fp64-free: the oracle is fp32/fp64 scalar C)
        pick = torch.randint(0, loc.shape[1], (1500,), generator=torch.Generator().manual_seed(3)).sort().values
        ref = c_oracle.forward(v.float().cpu().numpy(), shapes.cpu().numpy(), lsi.cpu().numpy(),
                               loc[:, pick].float().cpu().numpy(), attn[:, pick].float().cpu().numpy())
        err_o = relerr(a[:, pick.cuda()], torch.from_numpy(ref))
        MSDA.flags = 0
        t_mma = time_ms(lambda: MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, attn, 64))
        MSDA.flags = NO_MMA
        t_brick = time_ms(lambda: MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, attn, 64))
        MSDA.flags = 0
        print(json.dumps({"flagship": dist, "N": n, "mma_vs_brick": err, "mma_vs_oracle_1500q": err_o,
                          "ms_mma_median_min": t_mma, "ms_brick_median_min": t_brick}), flush=True)


if __name__ == "__main__":
    if "--flagship-only" not in sys.argv:
        small_cases()
    if "--quick" not in sys.argv:
        flagship(dists=tuple(sys.argv[sys.argv.index("--dists") + 1].split(",")) if "--dists" in sys.argv
                 else ("model", "model_nojitter", "uniform"))
