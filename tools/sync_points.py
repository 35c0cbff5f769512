#!/usr/bin/env python
"""Host synchronisations inside one eager training step (torch.cuda.set_sync_debug_mode("warn")): every one of them lets
the GPU's queue run dry.  Expected: none.    python tools/sync_points.py"""
import os
import sys
import warnings

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config  # noqa: E402
from transoar_amd.matcher import DenseTargets  # noqa: E402
from transoar_amd.train_step import TrainStep  # noqa: E402
from transoar_amd.transoarnet import TransoarNet, build_criterion  # noqa: E402

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
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as got:
    warnings.simplefilter("always")
    step(x, tg)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
syncs = [w for w in got if "synchroniz" in str(w.message).lower()]
print("synchronising calls in one eager step:", len(syncs))
for w in syncs[:20]:
    print("  %s:%d  %s" % (w.filename.split("repo/")[-1], w.lineno, str(w.message)[:120]))
