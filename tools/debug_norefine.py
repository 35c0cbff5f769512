import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, "miopen_db"))
from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config
from transoar_amd.matcher import DenseTargets
from transoar_amd.train_step import TrainStep
from transoar_amd.transoarnet import TransoarNet, build_criterion
dev = "cuda"
refine = len(sys.argv) > 1 and sys.argv[1] == "refine"
cfg = visceral_config(refine=refine, use_cuda=True)
cfg["bbox_properties"] = synthetic_bbox_properties(cfg["num_classes"], seed=0)
torch.manual_seed(0)
model = TransoarNet(cfg).to(dev)
step = TrainStep(model, build_criterion(cfg), cfg, amp_dtype=torch.bfloat16, graph=True)
g = torch.Generator(device=dev).manual_seed(1234)
x = torch.rand(2, 1, *cfg["volume_shape"], device=dev, generator=g)
targets = DenseTargets.from_list(synthetic_targets(2, cfg["num_classes"], seed=1, device=dev), cfg["num_classes"], dev)
t, _ = step(x, targets); print("eager0", float(t))
def nan_params(tag):
    bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    badg = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(tag, "nan params", bad[:5], len(bad), "nan grads", badg[:5], len(badg))
nan_params("after eager0")
step.capture(x, targets)
nan_params("after capture")
for i in range(4):
    t, losses = step(x, targets)
    torch.cuda.synchronize()
    print("replay", i, float(t), {k: round(float(v), 4) for k, v in list(losses.items())[:4]})
    nan_params("after replay %d" % i)
