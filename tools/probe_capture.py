"""Which pieces of the step survive HIP graph capture?"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, "miopen_db"))
import torch
import torch.nn as nn
import torch.nn.functional as F

def try_capture(name, fn):
    try:
        for _ in range(2): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay(); torch.cuda.synchronize()
        print("OK  ", name, flush=True)
    except Exception as e:
        print("FAIL", name, str(e).splitlines()[0][:100], flush=True)
        torch.cuda.synchronize()

dev = "cuda"
from tests import _inputs
from transoar_amd import MSDeformAttnFunction
from transoar_amd.backbone import EncoderCnnBlock
from transoar_amd.focused_decoder import FocusedDecoderLayer
from transoar_amd.config import visceral_config, synthetic_bbox_properties

# 1. MSDA fwd+bwd
value, shapes, lsi, loc, attn = _inputs.model_like_inputs(0, 1, [(8, 8, 8), (4, 4, 8)], device=dev)
v = value.bfloat16().requires_grad_(); lo = loc.clone().requires_grad_(); at = attn.clone().requires_grad_()
def f_msda():
    v.grad = lo.grad = at.grad = None
    MSDeformAttnFunction.apply(v, shapes, lsi, lo, at, 64).float().sum().backward()
try_capture("msda fwd+bwd", f_msda)

# 2. encoder block with HIP conv + IN (big enough for the HIP path) and MIOpen path
for name, (ci, co, s, D) in {"enc block hip (24->24 big)": (24, 24, 1, 64), "enc block miopen (96->192 s2)": (96, 192, 2, 16)}.items():
    blk = EncoderCnnBlock(ci, co, (3, 3, 3), (s, s, s)).to(dev)
    x = torch.randn(2, ci, D, D * 2, D * 2, device=dev, requires_grad=True)
    def f_blk(blk=blk, x=x):
        x.grad = None
        for p in blk.parameters(): p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            blk(x).float().sum().backward()
    try_capture(name, f_blk)

# 3. plain convs / conv transpose (FPN decoder pieces)
conv = nn.Conv3d(96, 384, 3, padding=1).to(dev); ct = nn.ConvTranspose3d(384, 192, 2, stride=2).to(dev); c1 = nn.Conv3d(192, 192, 1).to(dev)
xa = torch.randn(2, 96, 16, 16, 32, device=dev, requires_grad=True); xb = torch.randn(2, 384, 8, 8, 16, device=dev, requires_grad=True); xc = torch.randn(2, 192, 8, 8, 16, device=dev, requires_grad=True)
for name, (m, x) in {"conv3x3 96->384": (conv, xa), "convtranspose": (ct, xb), "conv1x1": (c1, xc)}.items():
    def f_c(m=m, x=x):
        x.grad = None
        for p in m.parameters(): p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            m(x).float().sum().backward()
    try_capture(name, f_c)

# 4. focused decoder layer
cfg = visceral_config(); props = synthetic_bbox_properties(20)
layer = FocusedDecoderLayer(384, 1024, 0.1, "relu", 8, cfg["neck"], props).to(dev)
tgt = torch.randn(2, 540, 384, device=dev, requires_grad=True); qp = torch.randn(2, 540, 384, device=dev)
src = torch.randn(2, 102400, 384, device=dev, requires_grad=True); sp = torch.randn(2, 102400, 384, device=dev)
def f_layer():
    tgt.grad = src.grad = None
    for p in layer.parameters(): p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        layer(tgt, qp, sp, src)[0].float().sum().backward()
try_capture("focused decoder layer", f_layer)

# 5. layernorm / linear / dropout on tokens
lin = nn.Sequential(nn.Linear(384, 1024), nn.ReLU(), nn.Dropout(0.1), nn.Linear(1024, 384), nn.LayerNorm(384)).to(dev)
xt = torch.randn(2, 20000, 384, device=dev, requires_grad=True)
def f_lin():
    xt.grad = None
    for p in lin.parameters(): p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lin(xt).float().sum().backward()
try_capture("ffn+ln+dropout", f_lin)
