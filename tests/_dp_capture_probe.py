"""Child process of tests/test_train_step_gpu.py::test_captured_data_parallel_step: a ONE-rank RCCL group
(TRANSOAR_FORCE_DP makes the gradient exchange active anyway), the training step captured WITH its exchange -- bucket
all-reduces launched by the gradient hooks inside the capture, the loss normalisers summed by the graph's first node,
the AdamW update behind the joins.  Prints one JSON line.

    python tests/_dp_capture_probe.py <port>
"""
import copy
import json
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ["TRANSOAR_FORCE_DP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from tests.test_train_step_gpu import _batch, _flagship  # noqa: E402
from transoar_amd.train_step import TrainStep, build_optimizer  # noqa: E402


def main():
    port = int(sys.argv[1])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    out = {}
    # ---- (a) captured gradients == eager gradients (exchange captured, optimizer eager so that replays leave the weights alone)
    cfg, model, crit = _flagship()
    step = TrainStep(model, crit, cfg, optimizer=build_optimizer(model, cfg), amp_dtype=torch.bfloat16, graph=True)
    assert step.reducer.active and step.capture_exchange and not step.capture_optimizer
    x, t = _batch(cfg, 1)
    params = {n: p for n, p in model.named_parameters() if p.requires_grad}
    model.zero_grad(set_to_none=True)
    # the eager reference pass runs on the stream the capture will use: autograd's AccumulateGrad nodes (and with them the
    # gradient hooks that launch RCCL) keep the stream of the first backward, and a hook that fires on the default stream
    # inside a capture on another stream takes the process down in hipStreamEndCapture (seen here: SIGSEGV)
    side = step.capture_stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step.reducer.begin()
        total, _ = step.loss(x, t)
        total.backward()
        step.reducer.finish()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    eager = {n: p.grad.detach().double().clone() for n, p in params.items() if p.grad is not None}
    step.capture(x, t, warmup=1)
    assert step._static_counts_local is not None, "the loss normalisers must be reduced inside the graph"
    worst = []
    for k in range(2):
        step._graph.replay()
        torch.cuda.synchronize()
        for n, ge in eager.items():
            gr = params[n].grad.double()
            worst.append((float((gr - ge).norm() / ge.norm().clamp_min(1e-30)), k, n))
    worst.sort(reverse=True)
    rest = [w for w in worst if not w[2].startswith("_backbone._encoder.")]
    out["grad_worst"], out["grad_worst_outside_encoder"] = worst[0], rest[0]
    out["n_buckets"] = len(step.reducer.buckets)
    del step, model

    # ---- (b) the whole step as one graph (exchange + AdamW inside): 3 replays move the weights like 3 eager steps of a twin
    cfg, model, crit = _flagship(clip=0.1)
    twin = copy.deepcopy(model)
    before = [p.detach().clone() for p in model.parameters()]
    cap = TrainStep(model, crit, cfg, amp_dtype=torch.bfloat16, graph=True)
    assert cap.capture_exchange and cap.capture_optimizer
    ref = TrainStep(twin, crit, cfg, amp_dtype=torch.bfloat16, graph=False)
    x, t = _batch(cfg, 1)
    cap.capture(x, t, warmup=2)
    out["capture_left_weights_alone"] = all(torch.equal(p, b) for p, b in zip(model.parameters(), before))
    for k in range(3):
        xb, tb = _batch(cfg, 10 + k)
        cap(xb, tb)
        ref(xb, tb)
    torch.cuda.synchronize()
    rel = []
    for (n, p), q, b in zip(model.named_parameters(), twin.parameters(), before):
        den = float((q.detach().double() - b.double()).norm())
        if den > 0:
            rel.append((float((p.detach().double() - q.detach().double()).norm()) / den, n))
    rel.sort(reverse=True)
    out["update_rel_worst"], out["update_rel_median"] = rel[0], rel[len(rel) // 2][0]
    out["update_rel_p90"] = rel[len(rel) // 10][0]
    out["loss_finite"] = bool(torch.isfinite(cap._static_total))
    print(json.dumps(out), flush=True)
    # the result is out: leave without the interpreter's teardown (a process group with captured collectives, its watchdog
    # thread and two HIP graphs going down together have aborted the process after the fact)
    try:
        torch.cuda.synchronize()
        dist.destroy_process_group()
    finally:
        import os
        sys.stdout.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
