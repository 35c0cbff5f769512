#!/usr/bin/env python
"""Localise differences between the point-column gather and the per-corner brick kernel on a small pyramid."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _inputs
from transoar_amd import MSDA

levels = [(8, 8, 16), (4, 4, 8), (2, 2, 4), (1, 1, 2)]
value, shapes, lsi, loc, attn = _inputs.model_like_inputs(0, 1, levels, device="cuda", jitter=0.3)
v = value.to(torch.bfloat16)
N, S, M, C = value.shape
L, P = 4, 4
starts = lsi.tolist() + [S]

def run(at, fl):
    MSDA.flags = fl
    return MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, at, 64).float().view(N, S, M, C)

for only_l in [None, 0, 1, 2, 3]:
    for only_p in [None, 0, 1, 2, 3]:
        if only_l is None and only_p is not None:
            continue
        at = attn.clone()
        if only_l is not None:
            mask = torch.zeros_like(at)
            if only_p is None:
                mask[:, :, :, only_l, :] = 1
            else:
                mask[:, :, :, only_l, only_p] = 1
            at = at * mask
        a, b = run(at, 0), run(at, 16)
        d = (a - b).abs()
        per_q_level = [d[:, starts[i]:starts[i + 1]].max().item() for i in range(4)]
        print("level", only_l, "point", only_p, "max diff by query level", ["%.3g" % x for x in per_q_level],
              "ref max %.3g" % b.abs().max().item(), "ch halves %.3g %.3g" % (d[..., :32].max().item(), d[..., 32:].max().item()),
              "heads", ["%.2g" % d[:, :, m].max().item() for m in range(M)])
# one query in detail
at = attn.clone()
a, b = run(at, 0), run(at, 16)
print("query 0 head 0 pcm ", a[0, 0, 0, :8].tolist())
print("query 0 head 0 ref ", b[0, 0, 0, :8].tolist())
print("ratio", (a[0, :6, 0, 0] / b[0, :6, 0, 0]).tolist())
d = (a - b).abs()[0]          # (S, M, C)
bad = d > 0.02
print("bad fraction", bad.float().mean().item())
D0, H0, W0 = levels[0]
badl0 = bad[:D0 * H0 * W0].view(D0, H0, W0, M, C)
print("by (d&1,h&1,w&1):", [[ (dd, hh, ww, round(badl0[dd::2, hh::2, ww::2].float().mean().item(), 3)) for ww in range(2)] for dd in range(2) for hh in range(2)])
print("by channel (first 64):", [round(x, 2) for x in badl0.float().mean((0, 1, 2, 3)).tolist()])
print("by head:", [round(x, 2) for x in badl0.float().mean((0, 1, 2, 4)).tolist()])
print("by w:", [round(x, 2) for x in badl0.float().mean((0, 1, 3, 4)).tolist()])
print("by d:", [round(x, 2) for x in badl0.float().mean((1, 2, 3, 4)).tolist()])
qi = 5
torch.set_printoptions(precision=3, linewidth=200)
print("pcm", a[0, qi, 0])
print("ref", b[0, qi, 0])
# does pcm channel c equal ref channel perm(c)?
ref_row = b[0, qi, 0]
for c in range(8, 24):
    j = (ref_row - a[0, qi, 0, c]).abs().argmin().item()
    print(c, "->", j, end="; ")
print()
