#!/usr/bin/env python
"""Does the captured training step reproduce the eager gradients on EVERY replay?  (tests/test_train_step_gpu.py as a
script, so that it can be run with ROCm's graph packet capture on:
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 TRANSOAR_TRUST_PACKET_CAPTURE=1 python tools/graph_replay_check.py [replays])
Prints per replay the number of parameters whose gradient norm left [1/4, 4] x the eager norm or is non-finite, and
the host time of a replay."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transoar_amd  # noqa: E402
from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config  # noqa: E402
from transoar_amd.matcher import DenseTargets  # noqa: E402
from transoar_amd.train_step import TrainStep  # noqa: E402
from transoar_amd.transoarnet import TransoarNet, build_criterion  # noqa: E402

replays = int(sys.argv[1]) if len(sys.argv) > 1 else 6
with_opt = "--opt" in sys.argv       # AdamW step after every replay, as the training step does
print("DEBUG_CLR_GRAPH_PACKET_CAPTURE =", os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"), "replay-safe:", transoar_amd.GRAPH_REPLAY_SAFE)
cfg = visceral_config(refine=True, use_cuda=True)
cfg["bbox_properties"] = synthetic_bbox_properties(cfg["num_classes"], seed=0)
torch.manual_seed(0)
model = TransoarNet(cfg)
with torch.no_grad():
    for p_ in model.parameters():        # the heads start at zero: no gradient would reach the body, the check would be blind
        if p_.dim() > 1 and float(p_.abs().max()) == 0:
            torch.nn.init.xavier_uniform_(p_)
model = model.cuda()
from transoar_amd.train_step import build_optimizer  # noqa: E402
step = TrainStep(model, build_criterion(cfg), cfg, optimizer=build_optimizer(model, cfg), amp_dtype=torch.bfloat16, graph=True)
x = torch.rand(2, 1, *cfg["volume_shape"], device="cuda", generator=torch.Generator(device="cuda").manual_seed(1234))
targets = DenseTargets.from_list(synthetic_targets(2, cfg["num_classes"], seed=1, device="cuda"), cfg["num_classes"], "cuda")
params = {n: p for n, p in model.named_parameters() if p.requires_grad}


def grad_norms():
    names = [n for n, p in params.items() if p.grad is not None]
    return dict(zip(names, torch.stack(torch._foreach_norm([params[n].grad for n in names])).float().cpu().tolist()))


step._eager_fwd_bwd(x, targets)
eager = grad_norms()
step.capture(x, targets, warmup=1)
for k in range(replays):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step._graph.replay()
    host_ms = (time.perf_counter() - t0) * 1e3
    got = grad_norms()
    if with_opt:
        step.optimizer.step()
        nonfinite = [n for n, p in params.items() if not torch.isfinite(p).all()]
        if nonfinite:
            print("  non-finite PARAMETERS after the optimizer step:", nonfinite[:5])
    bad = [n for n, rn in eager.items() if not (got.get(n) == got.get(n) and got.get(n, float("inf")) != float("inf")
                                                and (rn <= 1e-6 or 0.25 * rn <= got[n] <= 4.0 * rn))]
    probe = "_backbone._encoder._stages.0._block.0.weight"
    print("replay %d: host %.1f ms, loss %.6f, stem grad norm %.6e (eager %.6e), %d of %d gradients off %s" % (
        k, host_ms, float(step._static_total), got[probe], eager[probe], len(bad), len(eager), bad[:3]), flush=True)
