"""torch.profiler view of one training step: which aten ops launch the glue kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, "miopen_db"))
import torch
from torch.profiler import profile, ProfilerActivity
from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config
from transoar_amd.matcher import DenseTargets
from transoar_amd.train_step import TrainStep
from transoar_amd.transoarnet import TransoarNet, build_criterion
cfg = visceral_config(refine=True, use_cuda=True); cfg["bbox_properties"] = synthetic_bbox_properties(20)
torch.manual_seed(0)
model = TransoarNet(cfg).cuda(); step = TrainStep(model, build_criterion(cfg), cfg)
x = torch.rand(2, 1, 160, 160, 256, device="cuda")
tg = DenseTargets.from_list(synthetic_targets(2, 20, device="cuda"), 20, "cuda")
for _ in range(3): step(x, tg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(x, tg); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=48, max_shapes_column_width=70))
