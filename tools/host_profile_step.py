#!/usr/bin/env python
"""Host cost of one EAGER training step: cProfile over 5 steps with the GPU running behind
(one synchronize at the end), top functions by own time -> profiles/r04_host_profile.txt.  In graph mode the host cost is
one hipGraphLaunch; this is the figure that matters for the data-parallel default (eager, hooks, one process per GPU)."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config  # noqa: E402
from transoar_amd.matcher import DenseTargets  # noqa: E402
from transoar_amd.train_step import TrainStep  # noqa: E402
from transoar_amd.transoarnet import TransoarNet, build_criterion  # noqa: E402

cfg = visceral_config(refine=True, use_cuda=True)
cfg["bbox_properties"] = synthetic_bbox_properties(20)
torch.manual_seed(0)
model = TransoarNet(cfg).cuda()
step = TrainStep(model, build_criterion(cfg), cfg, graph=False)
x = torch.rand(2, 1, 160, 160, 256, device="cuda")
tg = DenseTargets.from_list(synthetic_targets(2, 20, device="cuda"), 20, "cuda")
for _ in range(5):
    step(x, tg)
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    step(x, tg)
t_enq = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / n
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step(x, tg)
pr.disable()
torch.cuda.synchronize()
out = io.StringIO()
out.write("eager step: host enqueue %.2f ms, wall %.2f ms per step (no profiler); below: cProfile of %d steps, own time\n" % (t_enq * 1e3, t_all * 1e3, n))
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28)
print(out.getvalue())
