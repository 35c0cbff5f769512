"""One eager + a few training steps of the AMOS-geometry model (3-level pyramid, 256x256x128 volume) on the GPU."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, "miopen_db"))
from transoar_amd.config import amos_config, synthetic_bbox_properties, synthetic_targets
from transoar_amd.matcher import DenseTargets
from transoar_amd.train_step import TrainStep
from transoar_amd.transoarnet import TransoarNet, build_criterion
dev = "cuda"
cfg = amos_config(refine=True, use_cuda=True)
cfg["bbox_properties"] = synthetic_bbox_properties(cfg["num_classes"], seed=0)
torch.manual_seed(0)
model = TransoarNet(cfg).to(dev)
step = TrainStep(model, build_criterion(cfg), cfg, amp_dtype=torch.bfloat16, graph=False)
x = torch.rand(1, 1, *cfg["volume_shape"], device=dev)
targets = DenseTargets.from_list(synthetic_targets(1, cfg["num_classes"], seed=1, device=dev), cfg["num_classes"], dev)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    total, _ = step(x, targets)
    torch.cuda.synchronize()
    print("step", i, "loss %.4f" % float(total), "%.1f ms" % ((time.perf_counter() - t0) * 1e3))
none = [n for n, p in model.named_parameters() if p.grad is None]
print("params without grad:", none)
assert all("q_proj" in n for n in none)
