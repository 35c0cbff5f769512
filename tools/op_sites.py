#!/usr/bin/env python
"""Which Python lines of this package issue aten operators in one eager training step, and how many: a TorchDispatchMode
counts every operator call with the nearest transoar_amd source line on the stack (operators the autograd engine runs in
the backward have no Python stack: they are booked under the autograd node that is running).  Debug aid for the launch
count of the step (~670 aten launches of ~5 us each).

    python tools/op_sites.py [--no-refine] [--top 60]
"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config  # noqa: E402
from transoar_amd.matcher import DenseTargets  # noqa: E402
from transoar_amd.train_step import TrainStep  # noqa: E402
from transoar_amd.transoarnet import TransoarNet, build_criterion  # noqa: E402

SKIP = {"aten.view.default", "aten._unsafe_view.default", "aten.t.default", "aten.transpose.int", "aten.permute.default",
        "aten.expand.default", "aten.slice.Tensor", "aten.select.int", "aten.unsqueeze.default", "aten.squeeze.dim",
        "aten.detach.default", "aten.alias.default", "aten.as_strided.default", "aten.reshape.default", "aten.split.Tensor",
        "aten.split_with_sizes.default", "aten.unbind.int", "aten.empty.memory_format", "aten.empty_like.default",
        "aten.empty_strided.default", "aten.new_empty.default", "aten.sym_size.int", "aten.lift_fresh.default"}


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.stats = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if name not in SKIP:
            where = "(autograd engine)"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "transoar_amd" in fr.filename:
                    where = "%s:%d" % (fr.filename.split("transoar_amd/")[-1], fr.lineno)
                    break
            self.stats[(where, name)] += 1
        return func(*args, **(kwargs or {}))


def main():
    cfg = visceral_config(refine="--no-refine" not in sys.argv, use_cuda=True)
    cfg["bbox_properties"] = synthetic_bbox_properties(20)
    torch.manual_seed(0)
    model = TransoarNet(cfg).cuda()
    step = TrainStep(model, build_criterion(cfg), cfg, graph=False)
    x = torch.rand(2, 1, 160, 160, 256, device="cuda")
    tg = DenseTargets.from_list(synthetic_targets(2, 20, device="cuda"), 20, "cuda")
    for _ in range(2):
        step(x, tg)
    torch.cuda.synchronize()
    with Count() as c:
        step(x, tg)
    torch.cuda.synchronize()
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 70
    by_file = collections.Counter()
    for (where, name), n in c.stats.items():
        by_file[where.split(":")[0]] += n
    print("operator calls (views and empties not counted):", sum(c.stats.values()))
    print("by file:", dict(by_file.most_common()))
    for (where, name), n in c.stats.most_common(top):
        print("%4d  %-36s %s" % (n, where, name))


if __name__ == "__main__":
    main()
