#!/usr/bin/env python
"""Captured step vs eager step on the SAME weights with dropout off, across optimizer steps: per-parameter relative L2
of the gradients.  Pinpoints which gradients a replay gets wrong (run with DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
TRANSOAR_TRUST_PACKET_CAPTURE=1 to examine ROCm's graph packet capture)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transoar_amd  # noqa: E402
from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config  # noqa: E402
from transoar_amd.matcher import DenseTargets  # noqa: E402
from transoar_amd.train_step import TrainStep  # noqa: E402
from transoar_amd.transoarnet import TransoarNet, build_criterion  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
print("packet capture:", os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"))
cfg = visceral_config(refine="--no-refine" not in sys.argv, use_cuda=True)
cfg["bbox_properties"] = synthetic_bbox_properties(cfg["num_classes"], seed=0)
torch.manual_seed(0)
model = TransoarNet(cfg)
with torch.no_grad():
    for p_ in model.parameters():
        if p_.dim() > 1 and float(p_.abs().max()) == 0:
            torch.nn.init.xavier_uniform_(p_)
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
    if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
        m.dropout = 0.0
model = model.cuda()
from transoar_amd.train_step import build_optimizer  # noqa: E402
step = TrainStep(model, build_criterion(cfg), cfg, optimizer=build_optimizer(model, cfg), amp_dtype=torch.bfloat16, graph=True)
x = torch.rand(2, 1, *cfg["volume_shape"], device="cuda", generator=torch.Generator(device="cuda").manual_seed(1234))
targets = DenseTargets.from_list(synthetic_targets(2, cfg["num_classes"], seed=1, device="cuda"), cfg["num_classes"], "cuda")
params = {n: p for n, p in model.named_parameters() if p.requires_grad}
step._eager_fwd_bwd(x, targets)
step.capture(x, targets, warmup=1)
for k in range(iters):
    step._graph.replay()
    g_graph = {n: p.grad.detach().clone() for n, p in params.items() if p.grad is not None}
    loss_g = float(step._static_total)
    loss_e = float(step._eager_fwd_bwd(step._static_x, step._static_t)[0])
    rel = []
    for n, g in g_graph.items():
        e = params[n].grad
        d = float((g - e).norm() / (e.norm() + 1e-20))
        rel.append((d if d == d else float("inf"), n))
    rel.sort(reverse=True)
    print("iter %d: loss graph %.6f eager %.6f; worst rel-L2: %s" % (k, loss_g, loss_e, [(round(a, 4), n[-60:]) for a, n in rel[:4]]), flush=True)
    if k == 1:
        for a, n in rel:
            if a > 0.2:
                g = g_graph[n]
                print("   off: %-70s rel %.3g  nan %d inf %d of %d  max|g| %.3g (eager %.3g)" % (
                    n, a, int(torch.isnan(g).sum()), int(torch.isinf(g).sum()), g.numel(), float(g[torch.isfinite(g)].abs().max()) if torch.isfinite(g).any() else float("nan"),
                    float(params[n].grad.abs().max())), flush=True)
    step.optimizer.step()
