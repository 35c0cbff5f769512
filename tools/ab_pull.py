import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests import _inputs
from transoar_amd import MSDA, _native
value, shapes, lsi, loc, attn = _inputs.model_like_inputs(0, 2, _inputs.VISCERAL_LEVELS, device="cuda")
v = value.bfloat16(); go = torch.randn(2, loc.shape[1], 384, device="cuda").bfloat16()
for flags in (0, 2, 0, 2):
    MSDA.flags = flags
    for _ in range(2): MSDA.ms_deform_attn_backward(v, shapes, lsi, loc, attn, go, 64)
    torch.cuda.synchronize(); _native.profile_enable(True); _native.profile_read()
    for _ in range(5): MSDA.ms_deform_attn_backward(v, shapes, lsi, loc, attn, go, 64)
    torch.cuda.synchronize(); p = _native.profile_read(); _native.profile_enable(False)
    print("flags", flags, {k: round(ms / n, 3) for k, (ms, n) in p.items() if n})
