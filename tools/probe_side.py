import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _inputs
from transoar_amd import MSDA
value, shapes, lsi, loc, attn = _inputs.model_like_inputs(0, 2, _inputs.VISCERAL_LEVELS, device="cuda")
v = value.to(torch.bfloat16); go = torch.randn(2, loc.shape[1], 384, device="cuda").to(torch.bfloat16)
def t(n=20):
    for _ in range(3): MSDA.ms_deform_attn_backward(v, shapes, lsi, loc, attn, go, 64)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): MSDA.ms_deform_attn_backward(v, shapes, lsi, loc, attn, go, 64)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for fl in (0, 8, 0, 8):   # 8 = TRANSOAR_MSDA3D_FORK
    MSDA.flags = fl
    print("flags", fl, "bwd ms %.3f" % t())
