#!/usr/bin/env python
"""Op-level micro-benchmark of the MSDeformAttn-3D kernels at the flagship
shape (N=2, S=Lq=117000, M=6, C=64, L=4, P=4).
Prints one JSON line per (distribution, dtype, direction).

    python tools/bench_msda.py [--iters 20] [--geometry visceral|amos] [--n 2]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _inputs  # noqa: E402  (input generators only; no oracle)
from transoar_amd import MSDA  # noqa: E402

HBM_PEAK = 8.0e12


def algorithmic_bytes(N, S, M, C, L, Lq, P, e, e_loc):
    """bytes one call must move if value is read once."""
    fwd = e * N * S * M * C + e * N * Lq * M * C + e_loc * N * Lq * M * L * P * 4
    # backward accumulates grad_value in fp32 for 16-bit storage
    e_gv = max(e, 4)
    bwd = e * (N * S * M * C + N * Lq * M * C) + 2 * e_gv * N * S * M * C + e_loc * 2 * N * Lq * M * L * P * 4
    return fwd, bwd


def time_ms(fn, iters, warmup=3):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--geometry", default="visceral")
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--dtypes", default="f32,bf16")
    ap.add_argument("--dists", default="model,uniform")
    ap.add_argument("--proj", action="store_true",
                    help="backward through transoar_msda3d_backward_proj (the training path of round 4: the sampling head's "
                         "backward folded into the query kernel, bf16 grad_proj instead of fp32 grad_loc / grad_attn)")
    args = ap.parse_args()
    levels = _inputs.VISCERAL_LEVELS if args.geometry == "visceral" else _inputs.AMOS_LEVELS
    for dist in args.dists.split(","):
        value, shapes, lsi, loc, attn = _inputs.model_like_inputs(0, args.n, levels, device="cuda")
        if dist == "uniform":   # ops/test.py distribution: no locality at all
            loc = torch.rand_like(loc)
        N, S, M, C = value.shape
        L, P, Lq = shapes.shape[0], loc.shape[4], loc.shape[1]
        for dt in args.dtypes.split(","):
            vdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16, "f64": torch.float64}[dt]
            ldt = torch.float64 if dt == "f64" else torch.float32
            v = value.to(vdt)
            lo, at = loc.to(ldt), attn.to(ldt)
            go = torch.randn(N, Lq, M * C, device="cuda").to(vdt)
            fwd_b, bwd_b = algorithmic_bytes(N, S, M, C, L, Lq, P, v.element_size(), lo.element_size())
            f_med, f_min = time_ms(lambda: MSDA.ms_deform_attn_forward(v, shapes, lsi, lo, at, 64), args.iters)
            bwd = MSDA.ms_deform_attn_backward_proj if (args.proj and dt != "f32" and dt != "f64") else MSDA.ms_deform_attn_backward
            b_med, b_min = time_ms(lambda: bwd(v, shapes, lsi, lo, at, go, 64), args.iters)
            for name, med, mn, nbytes in (("fwd", f_med, f_min, fwd_b), ("bwd(+zero+cast)", b_med, b_min, bwd_b)):
                print(json.dumps({"op": "msda3d_" + name, "dist": dist, "dtype": dt, "N": N, "S": S,
                                  "ms_median": round(med, 4), "ms_min": round(mn, 4),
                                  "algorithmic_MB": round(nbytes / 1e6, 1),
                                  "achieved_GBps": round(nbytes / med / 1e6, 1),
                                  "frac_of_8TBps": round(nbytes / (med * 1e-3) / HBM_PEAK, 4)}), flush=True)


if __name__ == "__main__":
    main()
