#!/usr/bin/env python
"""Device time of the aten operators of one eager training step, by operator and input shapes (torch.profiler): which of
the ~600 glue launches are worth fusing away.  Companion of op_sites.py (which gives the source lines).

    python tools/aten_time.py [--no-refine] [--top 50]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config  # noqa: E402
from transoar_amd.matcher import DenseTargets  # noqa: E402
from transoar_amd.train_step import TrainStep  # noqa: E402
from transoar_amd.transoarnet import TransoarNet, build_criterion  # noqa: E402


def main():
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 50
    cfg = visceral_config(refine="--no-refine" not in sys.argv, use_cuda=True)
    cfg["bbox_properties"] = synthetic_bbox_properties(20)
    torch.manual_seed(0)
    model = TransoarNet(cfg).cuda()
    step = TrainStep(model, build_criterion(cfg), cfg, graph=False)
    x = torch.rand(2, 1, 160, 160, 256, device="cuda")
    tg = DenseTargets.from_list(synthetic_targets(2, 20, device="cuda"), 20, "cuda")
    for _ in range(3):
        step(x, tg)
    torch.cuda.synchronize()
    n = 3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(n):
            step(x, tg)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        dev = getattr(e, "self_device_time_total", None)
        if dev is None:
            dev = e.self_cuda_time_total
        if dev > 0 and e.key.startswith("aten::"):
            rows.append((dev / n, e.count / n, e.key, str(e.input_shapes)[:150]))
    rows.sort(reverse=True)
    print("aten operators with device time, per step (%d steps): %.3f ms in %.0f calls" % (n, sum(r[0] for r in rows) / 1e3, sum(r[1] for r in rows)))
    for dev, cnt, key, shapes in rows[:top]:
        print("%8.1f us %6.1f calls  %-34s %s" % (dev, cnt, key, shapes))


if __name__ == "__main__":
    main()
