#!/usr/bin/env python
"""Does the training step train?  The flagship model on ONE fixed synthetic batch (2 volumes 160 x 160 x 256, 20 organs), AdamW
at the reference's learning rates (scripts/train.py:52-63), N steps: the total loss every 10 steps, for the eager step and for
the one-graph step from the same initial weights.    python tools/train_curve.py [steps]"""
import json
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config  # noqa: E402
from transoar_amd.matcher import DenseTargets  # noqa: E402
from transoar_amd.train_step import TrainStep  # noqa: E402
from transoar_amd.transoarnet import TransoarNet, build_criterion  # noqa: E402


def run(graph, steps):
    cfg = visceral_config(refine=True, use_cuda=True)
    cfg["bbox_properties"] = synthetic_bbox_properties(20)
    torch.manual_seed(0)
    model = TransoarNet(cfg).cuda()
    step = TrainStep(model, build_criterion(cfg), cfg, graph=graph)
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.rand(2, 1, 160, 160, 256, device="cuda", generator=g)
    tg = DenseTargets.from_list(synthetic_targets(2, 20, seed=1, device="cuda"), 20, "cuda")
    if graph:
        step.capture(x, tg)
    curve = []
    for i in range(steps):
        total, _ = step(x, tg)
        if i % 10 == 0 or i == steps - 1:
            curve.append((i, round(float(total), 4)))
    return curve


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    out = {"steps": steps, "eager": run(False, steps), "graph": run(True, steps)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
