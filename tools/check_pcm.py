#!/usr/bin/env python
"""Point-column gather (msda3d_pcm.hpp) against round 2's kernels on the flagship pyramid: differences and times.

    python tools/check_pcm.py [--geometry visceral|amos] [--n 2] [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _inputs  # noqa: E402
from transoar_amd import MSDA, tokens  # noqa: E402

Q32, NO_MMA = 32, 16


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
    args = ap.parse_args()
    levels = _inputs.VISCERAL_LEVELS if args.geometry == "visceral" else _inputs.AMOS_LEVELS
    vdt = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    for dist in args.dists.split(","):
        jitter = 0.0 if dist == "init" else 0.3
        value, shapes, lsi, loc, attn = _inputs.model_like_inputs(0, args.n, levels, device="cuda", jitter=jitter)
        if dist == "uniform":
            loc = torch.rand_like(loc)
        if dist == "wide":
            loc = (loc - 0.5) * 1.3 + 0.5 + 0.05 * torch.randn_like(loc)
        v = value.to(vdt)
        out = {}
        for name, fl in (("pcm", 0), ("q32", Q32), ("brick", NO_MMA)):
            MSDA.flags = fl
            out[name] = MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, attn, 64).float()
            ms = timed(lambda: MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, attn, 64), args.iters)
            print(json.dumps({"dist": dist, "kernel": name, "ms": round(ms, 4)}), flush=True)
        MSDA.flags = 0
        scale = out["brick"].abs().max().item()
        for a in ("pcm", "q32"):
            d = (out[a] - out["brick"]).abs().max().item()
            print(json.dumps({"dist": dist, "diff": a + " vs brick", "max_abs": d, "rel_to_max": d / scale}), flush=True)
        assert (out["pcm"] - out["brick"]).abs().max().item() <= 2.0 ** -7 * scale, "pcm differs"

    # fused head: proj -> (sampling_head -> gather) against the fused entry
    value, shapes, lsi, loc, attn = _inputs.model_like_inputs(0, args.n, levels, device="cuda")
    N, S, M, C = value.shape
    L, P = shapes.shape[0], 4
    g = torch.Generator(device="cuda").manual_seed(1)
    dirs = torch.tensor([(-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 1), (0, 1, 0), (1, 0, 0)], dtype=torch.float32, device="cuda")
    step = torch.arange(1, P + 1, dtype=torch.float32, device="cuda")
    off = (dirs[:, None, None, :] * step[None, None, :, None]).expand(M, L, P, 3)
    off = off + 0.6 * (torch.rand(N, S, M, L, P, 3, device="cuda", generator=g) - 0.5)
    logits = torch.randn(N, S, M, L * P, device="cuda", generator=g)
    proj = torch.cat((off.reshape(N, S, -1), logits.reshape(N, S, -1)), -1).to(torch.bfloat16).contiguous()
    ref = _inputs.reference_points(shapes.cpu()).to("cuda")[:, :, None, :].expand(1, S, L, 3).contiguous()
    v = value.to(vdt)
    lo2, at2 = tokens.sampling_head(proj, ref, shapes, M, L, P)
    two = MSDA.ms_deform_attn_forward(v, shapes, lsi, lo2, at2, 64).float()
    fused = MSDA.ms_deform_attn_forward_fused(v, shapes, proj, ref).float()
    d = (two - fused).abs().max().item()
    print(json.dumps({"fused vs head+gather": d, "rel_to_max": d / two.abs().max().item()}), flush=True)
    ms_f = timed(lambda: MSDA.ms_deform_attn_forward_fused(v, shapes, proj, ref), args.iters)
    ms_h = timed(lambda: tokens.sampling_head(proj, ref, shapes, M, L, P), args.iters)
    ms_g = timed(lambda: MSDA.ms_deform_attn_forward(v, shapes, lsi, lo2, at2, 64), args.iters)
    e = v.element_size()
    b_fused = e * N * S * M * C * 2 + 2 * N * S * 4 * M * L * P + 4 * S * L * 3
    print(json.dumps({"fused_ms": round(ms_f, 4), "head_ms": round(ms_h, 4), "gather_ms": round(ms_g, 4),
                      "fused_algorithmic_MB": round(b_fused / 1e6, 1),
                      "fused_frac_of_8TBps": round(b_fused / (ms_f * 1e-3) / 8e12, 4)}), flush=True)
    assert d <= 2.0 ** -7 * two.abs().max().item(), "fused differs"


if __name__ == "__main__":
    main()
