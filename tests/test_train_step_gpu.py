"""The captured (HIP graph) training step must compute what the eager step computes -- on EVERY replay.

ROCm 7.x's graph packet capture made the second and later replays of the refine-on step produce
inf/garbage weight gradients while the first one was exact; the package switches
it off (transoar_amd/__init__.py).  This test replays the captured forward+loss+backward without
optimizer steps in between and compares gradients between replays and against an eager pass: weights and
inputs are identical, only the dropout masks differ.
"""
import pytest
import torch


@pytest.mark.gpu
def test_captured_step_replays_reproduce_eager_gradients():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import transoar_amd
    from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config
    from transoar_amd.matcher import DenseTargets
    from transoar_amd.train_step import TrainStep
    from transoar_amd.transoarnet import TransoarNet, build_criterion
    assert transoar_amd.GRAPH_REPLAY_SAFE, "tests/conftest.py must switch graph packet capture off before HIP starts"
    cfg = visceral_config(refine=True, use_cuda=True)
    cfg["bbox_properties"] = synthetic_bbox_properties(cfg["num_classes"], seed=0)
    torch.manual_seed(0)
    model = TransoarNet(cfg)
    with torch.no_grad():
        for p_ in model.parameters():    # the heads start at zero: no gradient would reach the body and the check would be blind
            if p_.dim() > 1 and float(p_.abs().max()) == 0:
                torch.nn.init.xavier_uniform_(p_)
    model = model.cuda()
    # an explicit (eager) optimizer: the captured graph then holds forward + loss + backward only and replays
    # leave the weights alone, so that every replay can be compared with the eager pass (the default on one GPU
    # captures the AdamW update as well)
    from transoar_amd.train_step import build_optimizer
    step = TrainStep(model, build_criterion(cfg), cfg, optimizer=build_optimizer(model, cfg), amp_dtype=torch.bfloat16, graph=True)
    assert not step.capture_optimizer
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.rand(2, 1, *cfg["volume_shape"], device="cuda", generator=g)
    targets = DenseTargets.from_list(synthetic_targets(2, cfg["num_classes"], seed=1, device="cuda"),
                                     cfg["num_classes"], "cuda")
    params = {n: p for n, p in model.named_parameters() if p.requires_grad}

    def grad_norms():
        """{name: L2 norm of .grad} with one host sync (inf/nan gradients give inf/nan norms)."""
        names = [n for n, p in params.items() if p.grad is not None]
        norms = torch.stack(torch._foreach_norm([params[n].grad for n in names])).float().cpu().tolist()
        return dict(zip(names, norms))

    step._eager_fwd_bwd(x, targets)
    eager = grad_norms()
    step.capture(x, targets, warmup=1)
    replays = []
    for _ in range(3):
        step._graph.replay()
        replays.append(grad_norms())
    total = float(step._static_total)
    assert total == total, "captured step returns a NaN loss"
    bad = []
    for k, got in enumerate(replays):
        for n, rn in eager.items():
            gn = got.get(n)
            if gn is None:
                bad.append((k, n, "no gradient"))
            elif gn != gn or gn == float("inf"):
                bad.append((k, n, "non-finite"))
            # dropout changes individual gradients by tens of percent; garbage changes them by orders of magnitude
            elif rn > 1e-6 and not (0.25 * rn <= gn <= 4.0 * rn):
                bad.append((k, n, rn, gn))
    assert not bad, bad[:8]

    # ---- the loss normalisers must follow the batch, not the captured one (they are fed through a static device
    # buffer): replay with half of the classes absent and compare with the eager losses of the same batch
    half = synthetic_targets(2, cfg["num_classes"], seed=1, device="cuda")
    for t in half:
        keep = t["labels"] % 2 == 0
        t["labels"], t["boxes"] = t["labels"][keep], t["boxes"][keep]
    t2 = DenseTargets.from_list(half, cfg["num_classes"], "cuda")
    assert t2.num_boxes * 2 == targets.num_boxes
    step._static_t.boxes.copy_(t2.boxes)
    step._static_t.present.copy_(t2.present)
    step._static_t.num_boxes = t2.num_boxes
    step._static_counts.copy_(step._local_counts(step._static_t))
    step._graph.replay()
    got = {k: float(v) for k, v in step._static_losses.items()}
    model.zero_grad(set_to_none=False)
    with torch.no_grad():
        want = {k: float(v) for k, v in step.loss(x, t2)[1].items()}
    for k in ("bbox", "giou", "cls", "bbox_0", "giou_1"):
        assert abs(got[k] - want[k]) <= 0.25 * abs(want[k]) + 1e-4, (k, got[k], want[k])    # dropout noise << the factor 2 of a frozen count
