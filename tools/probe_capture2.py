import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, "miopen_db"))
import torch

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
        try: torch.cuda.synchronize()
        except Exception: pass

from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config
from transoar_amd.matcher import DenseTargets
from transoar_amd.transoarnet import TransoarNet, build_criterion
from transoar_amd.backbone import EncoderCnnBlock
dev = "cuda"
cfg = visceral_config(refine=True, use_cuda=True); cfg["bbox_properties"] = synthetic_bbox_properties(20)
torch.manual_seed(0)
model = TransoarNet(cfg).to(dev).train(); crit = build_criterion(cfg)
x = torch.rand(2, 1, 160, 160, 256, device=dev)
tg = DenseTargets.from_list(synthetic_targets(2, 20, device=dev), 20, dev)
def zero():
    for p in model.parameters(): p.grad = None

# stage 0 block alone (Cin = 1)
blk0 = model._backbone._encoder._stages[0]
def f_b0():
    zero()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        blk0(x).float().sum().backward()
try_capture("stage0 block (Cin=1)", f_b0)

def f_enc():
    zero()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model._backbone._encoder(x)
        sum(o.float().sum() for o in out.values()).backward()
try_capture("encoder", f_enc)

def f_bb():
    zero()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model._backbone(x)
        sum(o.float().sum() for o in out.values()).backward()
try_capture("backbone (encoder+fpn+refine)", f_bb)

def f_model():
    zero()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(x)
        (out["pred_logits"].float().sum() + out["pred_boxes"].float().sum()).backward()
try_capture("whole model, simple loss", f_model)

def f_full():
    zero()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(x)
        losses = crit(out, tg, None, model._anchors)
        sum(losses.values()).backward()
try_capture("whole model + criterion", f_full)
