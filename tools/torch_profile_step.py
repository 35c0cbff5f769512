"""torch.profiler view of one eager training step: which aten ops -- and which Python lines -- launch the glue
kernels.  Writes gpurun_out/step_ops.txt (by op and input shape) and gpurun_out/step_stacks.txt (by source line)."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, "tools", "miopen_db"))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config  # noqa: E402
from transoar_amd.matcher import DenseTargets  # noqa: E402
from transoar_amd.train_step import TrainStep  # noqa: E402
from transoar_amd.transoarnet import TransoarNet, build_criterion  # noqa: E402

cfg = visceral_config(refine="--no-refine" not in sys.argv, use_cuda=True, swin="--swin" in sys.argv)
cfg["bbox_properties"] = synthetic_bbox_properties(20)
torch.manual_seed(0)
model = TransoarNet(cfg).cuda()
step = TrainStep(model, build_criterion(cfg), cfg, graph=False)
x = torch.rand(2, 1, 160, 160, 256, device="cuda")
tg = DenseTargets.from_list(synthetic_targets(2, 20, device="cuda"), 20, "cuda")
for _ in range(3):
    step(x, tg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(x, tg)
    torch.cuda.synchronize()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "step_ops.txt"), "w") as f:
    f.write(prof.key_averages(group_by_input_shape=True).table(
        sort_by="self_cuda_time_total", row_limit=80, max_name_column_width=48, max_shapes_column_width=70))

# every (op, input shapes) with its own GPU time, no row limit: gpurun_out/step_ops_all.tsv
with open(os.path.join(ROOT, "gpurun_out", "step_ops_all.tsv"), "w") as f:
    for e in sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.self_device_time_total):
        if e.self_device_time_total > 0:
            f.write("%s\t%d\t%.1f\t%s\n" % (e.key, e.count, e.self_device_time_total, str(e.input_shapes)[:200]))

# kernels per source line of this package (innermost transoar_amd frame of the launching op)
by_line = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
        continue
    where = "?"
    for fr in ev.stack or []:
        if "transoar_amd" in fr and "site-packages" not in fr:
            where = fr.split("transoar_amd/")[-1]
            break
    n = len(ev.kernels)
    t = sum(k.duration for k in ev.kernels)
    a = by_line[(where, ev.name)]
    a[0] += n
    a[1] += t
rows = sorted(by_line.items(), key=lambda kv: -kv[1][0])
with open(os.path.join(ROOT, "gpurun_out", "step_stacks.txt"), "w") as f:
    f.write("launches  gpu_us  where  op\n")
    for (where, name), (n, t) in rows[:150]:
        f.write("%5d %9.1f  %-60s %s\n" % (n, t, where[:60], name))
    f.write("total launches %d, gpu ms %.2f\n" % (sum(v[0] for v in by_line.values()), sum(v[1] for v in by_line.values()) / 1e3))
print(open(os.path.join(ROOT, "gpurun_out", "step_stacks.txt")).read()[:6000])
