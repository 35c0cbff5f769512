"""Which Python lines of this package make copies in one eager training step: Tensor.contiguous / .to / .float / .clone /
torch.cat calls that return new memory, by source line and bytes (debug aid, not part of the product)."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, "tools", "miopen_db"))
import torch
from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config
from transoar_amd.matcher import DenseTargets
from transoar_amd.train_step import TrainStep
from transoar_amd.transoarnet import TransoarNet, build_criterion
cfg = visceral_config(refine="--no-refine" not in sys.argv, use_cuda=True, swin="--swin" in sys.argv); cfg["bbox_properties"] = synthetic_bbox_properties(20)
torch.manual_seed(0)
model = TransoarNet(cfg).cuda(); step = TrainStep(model, build_criterion(cfg), cfg, graph=False)
x = torch.rand(2, 1, 160, 160, 256, device="cuda")
tg = DenseTargets.from_list(synthetic_targets(2, 20, device="cuda"), 20, "cuda")
for _ in range(3): step(x, tg)
torch.cuda.synchronize()
stats = collections.defaultdict(lambda: [0, 0])
def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "transoar_amd" in fr.filename:
            return "%s:%d" % (fr.filename.split("transoar_amd/")[-1], fr.lineno)
    return "?"
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        out = orig(self, *a, **k)
        if isinstance(out, torch.Tensor) and out.is_cuda and (out.data_ptr() != self.data_ptr() or out.dtype != self.dtype):
            s = stats[(name, where())]; s[0] += 1; s[1] += out.numel() * out.element_size()
        return out
    setattr(torch.Tensor, name, f)
for n in ("contiguous", "to", "float", "clone", "bfloat16", "reshape", "flatten"): wrap(n)
ocat = torch.cat
def cat(ts, *a, **k):
    out = ocat(ts, *a, **k)
    if out.is_cuda:
        s = stats[("cat", where())]; s[0] += 1; s[1] += out.numel() * out.element_size()
    return out
torch.cat = cat
step(x, tg); torch.cuda.synchronize()
rows = sorted(stats.items(), key=lambda kv: -kv[1][1])
for (n, w), (c, b) in rows[:60]:
    print("%4d calls %9.1f MB  %-11s %s" % (c, b / 1e6, n, w))
