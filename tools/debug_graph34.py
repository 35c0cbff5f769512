"""Which half of the MSDeformAttn op is involved in the corruption at the 34th graph replay?
   python tools/debug_graph34.py {none|skip_bwd|skip_bwd_loc|skip_bwd_value}   (dev tool)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, "miopen_db"))
import transoar_amd.ms_deform_attn as mda
from transoar_amd import msda as MSDA
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
orig = MSDA.ms_deform_attn_backward
def patched(value, shapes, starts, loc, attn, go, step):
    if mode == "skip_bwd":
        return [torch.zeros_like(value), torch.zeros_like(loc), torch.zeros_like(attn)]
    gv, gl, ga = orig(value, shapes, starts, loc, attn, go, step)
    if mode == "skip_bwd_loc":
        return [gv, torch.zeros_like(gl), torch.zeros_like(ga)]
    if mode == "skip_bwd_value":
        return [torch.zeros_like(gv), gl, ga]
    return [gv, gl, ga]
mda.MSDA.ms_deform_attn_backward = patched
from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config
from transoar_amd.matcher import DenseTargets
from transoar_amd.train_step import TrainStep
from transoar_amd.transoarnet import TransoarNet, build_criterion
if mode == "rocblas":
    torch.backends.cuda.preferred_blas_library("cublas")      # rocBLAS instead of hipBLASLt
cfg = visceral_config(refine=True, use_cuda=True)
cfg["bbox_properties"] = synthetic_bbox_properties(cfg["num_classes"], seed=0)
torch.manual_seed(0)
model = TransoarNet(cfg).cuda()
if mode == "no_dropout":
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
if mode == "no_refine_dropout":
    for m in model._backbone._decoder._refine.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
step = TrainStep(model, build_criterion(cfg), cfg, amp_dtype=torch.bfloat16, graph=True)
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.rand(2, 1, *cfg["volume_shape"], device="cuda", generator=g)
targets = DenseTargets.from_list(synthetic_targets(2, cfg["num_classes"], seed=1, device="cuda"), cfg["num_classes"], "cuda")
step(x, targets)
step.capture(x, targets)
names = ["_backbone._decoder._refine.level_embed", "_backbone._decoder._refine.refine_def_attn.layers.0.linear1.weight",
         "_backbone._decoder._refine.refine_def_attn.layers.1.norm2.weight", "_backbone._encoder._stages.0._block.0.weight",
         "_neck.decoder.layers.0.linear1.weight", "_backbone._decoder._out.0.weight"]
params = dict(model.named_parameters())
for i in range(45):
    t, _ = step(x, targets)
    torch.cuda.synchronize()
    v = float(t)
    if mode == "gradnorm" and (i < 6 or i > 28):
        print(i + 1, "loss %.4f" % v, ["%.3e" % float(params[n].grad.float().norm()) for n in names], ["%.3e" % float(params[n].float().abs().max()) for n in names[:3]], flush=True)
    if v != v:
        print(mode, "NaN at replay", i + 1); break
else:
    print(mode, "45 replays fine, last loss %.4f" % v)
