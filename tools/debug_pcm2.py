import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _inputs
from transoar_amd import MSDA
torch.set_printoptions(precision=3, linewidth=220)
levels = [(8, 8, 16), (4, 4, 8), (2, 2, 4), (1, 1, 2)]
value, shapes, lsi, loc, attn = _inputs.model_like_inputs(0, 1, levels, device="cuda", jitter=0.3)
N, S, M, C = value.shape
def run(v, fl, lo=loc):
    MSDA.flags = fl
    return MSDA.ms_deform_attn_forward(v.to(torch.bfloat16), shapes, lsi, lo, attn, 64).float().view(N, S, M, C)
# E1: all ones
v1 = torch.ones_like(value)
a, b = run(v1, 0), run(v1, 16)
print("E1 ones: pcm q100", a[0, 100, 0, :20], "ref", b[0, 100, 0, :4])
# E2: value = row index / 100 for all channels
rows = torch.arange(S, device="cuda", dtype=torch.float32)[None, :, None, None].expand(N, S, M, C) / 64
a, b = run(rows.contiguous(), 0), run(rows.contiguous(), 16)
print("E2 rowid: pcm q100", a[0, 100, 0, :20], "ref", b[0, 100, 0, :4])
# E3: only channel c nonzero
for c in (0, 8, 9, 16, 40):
    v3 = torch.zeros_like(value); v3[..., c] = value[..., c]
    a, b = run(v3, 0), run(v3, 16)
    nz = (a[0, 100, 0].abs() > 1e-6).nonzero().flatten().tolist()
    print("E3 channel", c, "pcm nonzero channels", nz, "pcm", a[0, 100, 0, c].item(), "ref", b[0, 100, 0, c].item())
