#!/usr/bin/env python
"""Per-layer timing of the hand-written conv3d kernels vs PyTorch/MIOpen (tuned
find-db in miopen_db/) on the backbone's real shapes, batch 2, bf16."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, "tools", "miopen_db"))
import torch
import torch.nn.functional as F
from transoar_amd import conv3d as C

LAYERS = [  # name, Cin, Cout, D, H, W (input), stride
    ("s0c1", 1, 24, 160, 160, 256, 1), ("s0c2", 24, 24, 160, 160, 256, 1), ("s1c1", 24, 48, 160, 160, 256, 2),
    ("s1c2", 48, 48, 80, 80, 128, 1), ("s2c1", 48, 96, 80, 80, 128, 2), ("s2c2", 96, 96, 40, 40, 64, 1),
    ("s3c1", 96, 192, 40, 40, 64, 2), ("s3c2", 192, 192, 20, 20, 32, 1), ("s4c1", 192, 384, 20, 20, 32, 2),
    ("s4c2", 384, 384, 10, 10, 16, 1), ("s5c1", 384, 768, 10, 10, 16, 2), ("s5c2", 768, 768, 5, 5, 8, 1),
    ("outP2", 96, 384, 40, 40, 64, 1), ("outP3", 192, 384, 20, 20, 32, 1), ("outP4", 384, 384, 10, 10, 16, 1),
]

def t_ms(fn, n=5):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))[n // 2]

only = sys.argv[1:] 
for name, ci, co, d, h, w, s in LAYERS:
    if only and name not in only: continue
    x = torch.randn(2, ci, d, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    wt = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
    res = {"layer": name, "shape": [ci, co, d, h, w, s]}
    if ci > 1:
        wk = C._pack_taps(wt)
        y = C.conv3d_k3_forward(x, wk, None, s)
        res["hip_fwd"] = round(t_ms(lambda: C.conv3d_k3_forward(x, wk, None, s)), 3)
        gy = torch.randn_like(y)
        wtr = wt.flip(2, 3, 4).permute(2, 3, 4, 1, 0).reshape(27, ci, co).to(torch.bfloat16).contiguous()
        res["hip_dgrad"] = round(t_ms(lambda: C.conv3d_k3_forward(gy, wtr, None, 1, dilated_input=(s == 2))), 3)
        if s == 1 and C.lds_wgrad_supported(x, gy):
            res["hip_wgrad_lds"] = round(t_ms(lambda: C.conv3d_k3_wgrad_lds(x, gy)), 3)     # what the model runs
        else:
            res["hip_wgrad_v1"] = round(t_ms(lambda: C.conv3d_k3_wgrad(x, gy, s)), 3)      # off by default
    else:
        conv = C.Conv3dK3(ci, co, 3, padding=1, bias=False).cuda()
        res["hip_fwd"] = round(t_ms(lambda: conv(x)), 3)
        y = conv(x); gy = torch.randn_like(y)
        xc = x.contiguous()
        res["hip_wgrad_c1"] = round(t_ms(lambda: C.conv3d_c1_wgrad(xc, gy)), 3)
    flop = 2 * 27 * ci * co * y.shape[2] * y.shape[3] * y.shape[4] * 2
    res["fwd_TFs"] = round(flop / res["hip_fwd"] / 1e9, 1)
    # MIOpen (NCDHW bf16 operands)
    xn = x.contiguous().requires_grad_(ci > 1); wn = wt.to(torch.bfloat16).requires_grad_()
    res["miopen_fwd"] = round(t_ms(lambda: F.conv3d(xn, wn, stride=s, padding=1)), 3)
    yn = F.conv3d(xn, wn, stride=s, padding=1); gn = torch.randn_like(yn)
    ins = (xn, wn) if ci > 1 else (wn,)
    res["miopen_bwd"] = round(t_ms(lambda: torch.autograd.grad(yn, ins, gn, retain_graph=True)), 3)
    print(json.dumps(res), flush=True)
    del x, y, gy, xn, yn, gn
