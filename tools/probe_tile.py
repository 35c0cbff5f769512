#!/usr/bin/env python
"""Per-kernel times of the MSDA backward under TRANSOAR_DBG switches (dev tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _inputs
from transoar_amd import MSDA, _native
value, shapes, lsi, loc, attn = _inputs.model_like_inputs(0, 2, _inputs.VISCERAL_LEVELS, device="cuda")
print("shapes", shapes.tolist(), "loc", tuple(loc.shape))
v = value.to(torch.bfloat16); go = torch.randn(v.shape[0], loc.shape[1], v.shape[2] * v.shape[3], device="cuda").to(torch.bfloat16)
names = ["fwd", "bwd_query", "cell_count", "scan", "cell_fill", "pull", "fwd_generic", "bwd_generic"]
for dbg in sys.argv[1:] or ["0"]:
    if dbg != "0": os.environ["TRANSOAR_DBG"] = dbg
    for _ in range(2): MSDA.ms_deform_attn_backward(v, shapes, lsi, loc, attn, go, 64)
    torch.cuda.synchronize()
    _native.profile_enable(True)
    for _ in range(5): MSDA.ms_deform_attn_backward(v, shapes, lsi, loc, attn, go, 64)
    torch.cuda.synchronize()
    res = _native.profile_read()
    _native.profile_enable(False)
    print("dbg", dbg, {k: round(m / c, 3) for k, (m, c) in res.items() if c})
