"""Which parameters of the flagship model receive a (non-zero) gradient in one training step?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, "tools", "miopen_db"))
from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config
from transoar_amd.matcher import DenseTargets
from transoar_amd.train_step import TrainStep
from transoar_amd.transoarnet import TransoarNet, build_criterion
dev = "cuda"
cfg = visceral_config(refine=True, use_cuda=True)
cfg["bbox_properties"] = synthetic_bbox_properties(cfg["num_classes"], seed=0)
torch.manual_seed(0)
model = TransoarNet(cfg).to(dev)
step = TrainStep(model, build_criterion(cfg), cfg, amp_dtype=torch.bfloat16, graph=False)
x = torch.rand(2, 1, *cfg["volume_shape"], device=dev)
targets = DenseTargets.from_list(synthetic_targets(2, cfg["num_classes"], seed=1, device=dev), cfg["num_classes"], dev)
step(x, targets)            # one optimizer step first: the heads' last layers start at zero, so the very
step(x, targets)            # first backward carries an all-zero gradient into the rest of the network
for p in model.parameters():
    p.grad = None
total, losses = step.loss(x, targets)
total.backward()
none, zero, ok = [], [], []
for n, p in model.named_parameters():
    if p.grad is None: none.append(n)
    elif float(p.grad.abs().sum()) == 0.0: zero.append(n)
    else: ok.append(n)
print("with grad", len(ok), "zero grad", len(zero), "no grad", len(none))
print("NONE:", [n for n in none][:40])
print("ZERO:", [n for n in zero][:40])
