"""The fused set criterion (csrc/criterion.hip, include/transoar_criterion.h: one forward and one backward launch) against the
torch mirror of transoar/models/criterion.py + matcher.py + utils/bboxes.py (transoar_amd/criterion.py, matcher.py, bboxes.py --
itself pinned to the reference by goldens g7 / g10 / g11, which run through the fused kernel on the GPU as well).

Tolerances: every loss 2e-6 relative (fp32 sums of ~1000 terms in another order), gradients 1e-5 of the tensor maximum, the
matches identical (the test inputs have no tied costs)."""
import pytest
import torch


def _case(layers, n, organs, r, logits_dtype, seed, absent=3):
    g = torch.Generator(device="cuda").manual_seed(seed)
    q = organs * r
    logits = (torch.randn(layers, n, q, 1, device="cuda", generator=g) * 2).to(logits_dtype)
    centre = torch.rand(layers, n, q, 3, device="cuda", generator=g) * 0.8 + 0.1
    size = torch.rand(layers, n, q, 3, device="cuda", generator=g) * 0.3 + 0.02
    boxes = torch.cat((centre, size), -1)
    boxes[-1, 0, 5, 0] = -0.01                       # a centre the clamp cuts: its gradient must stop there
    anchors = torch.cat((torch.rand(q, 3, device="cuda", generator=g) * 0.8 + 0.1, torch.rand(q, 3, device="cuda", generator=g) * 0.3 + 0.05), -1)
    tgt = torch.cat((torch.rand(n, organs, 3, device="cuda", generator=g) * 0.6 + 0.2, torch.rand(n, organs, 3, device="cuda", generator=g) * 0.3 + 0.05), -1)
    present = torch.ones(n, organs, dtype=torch.bool, device="cuda")
    for k in range(absent):
        present[k % n, (5 * k + 1) % organs] = False
    tgt = tgt * present[..., None]
    return logits, boxes, anchors, tgt, present


def _run(crit, logits, boxes, anchors, tgt, present, counts, fused):
    from transoar_amd import fused_criterion
    from transoar_amd.matcher import DenseTargets
    lg = logits.clone().requires_grad_(True)
    bx = boxes.clone().requires_grad_(True)
    out = {"pred_logits": lg[-1], "pred_boxes": bx[-1],
           "aux_outputs": [{"pred_logits": a, "pred_boxes": b} for a, b in zip(lg[:-1], bx[:-1])]}
    if counts == "device":
        c = torch.stack((present.sum().float() + 3, present.sum().float() + 2))      # as if summed over data-parallel ranks
        targets = DenseTargets(tgt, present, c[0], c[1])
    else:
        targets = DenseTargets(tgt, present, int(present.sum()))
    was = fused_criterion.ENABLED
    fused_criterion.ENABLED = fused
    try:
        losses = crit(out, targets, None, anchors)
    finally:
        fused_criterion.ENABLED = was
    assert (getattr(losses, "vector", None) is not None) == fused
    w = torch.linspace(0.5, 2.0, len(losses), device="cuda")
    total = sum(wi * v for wi, v in zip(w, losses.values()))
    total.backward()
    return {k: float(v.detach()) for k, v in losses.items()}, lg.grad.float(), bx.grad.float()


@pytest.mark.gpu
@pytest.mark.parametrize("layers,n,organs,r,dtype,costs,counts", [
    (4, 2, 20, 27, torch.bfloat16, (1, 0, 0), "host"),       # the flagship's form (config.py: class cost only)
    (4, 2, 20, 27, torch.float32, (1, 0, 0), "device"),
    (1, 3, 5, 7, torch.float32, (2, 5, 2), "host"),           # the reference's default costs, no auxiliary outputs
    (6, 1, 20, 27, torch.bfloat16, (1, 1, 1), "device"),
    (2, 2, 4, 64, torch.float32, (1, 0, 2), "host"),          # the widest class the kernel takes
])
def test_fused_criterion_matches_the_torch_mirror(layers, n, organs, r, dtype, costs, counts):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from transoar_amd.criterion import TransoarCriterion
    from transoar_amd.matcher import Matcher
    crit = TransoarCriterion(organs, Matcher(*costs, anchor_matching=True, num_organs=organs), seg_proxy=False, seg_fg_bg=False)
    case = _case(layers, n, organs, r, dtype, seed=layers * 100 + r)
    ref, ref_gl, ref_gb = _run(crit, *case, counts, fused=False)
    got, got_gl, got_gb = _run(crit, *case, counts, fused=True)
    assert list(got) == list(ref)
    for k in ref:
        assert abs(got[k] - ref[k]) <= 2e-6 * abs(ref[k]) + 1e-7, (k, got[k], ref[k])
    assert got["segce"] == 0 and got["segdice"] == 0
    for a, b, name in ((got_gl, ref_gl, "logits"), (got_gb, ref_gb, "boxes")):
        assert torch.isfinite(a).all(), name
        tol = (3e-3 if dtype == torch.bfloat16 and name == "logits" else 1e-5) * float(b.abs().max())     # bf16 logits: the gradient is rounded to bf16
        assert float((a - b).abs().max()) <= tol, (name, float((a - b).abs().max()), float(b.abs().max()))
    assert float(got_gl[:-1].abs().max()) == 0 if layers > 1 else True        # auxiliary logits only steer the matching


@pytest.mark.gpu
def test_fused_criterion_declines_what_it_does_not_cover():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from transoar_amd import fused_criterion
    from transoar_amd.criterion import TransoarCriterion
    from transoar_amd.matcher import DenseTargets, Matcher
    logits, boxes, anchors, tgt, present = _case(2, 2, 4, 5, torch.float32, seed=3, absent=1)
    out = {"pred_logits": logits[-1], "pred_boxes": boxes[-1], "aux_outputs": [{"pred_logits": logits[0], "pred_boxes": boxes[0]}]}
    targets = DenseTargets(tgt, present, int(present.sum()))
    boxes_matching = TransoarCriterion(4, Matcher(1, 0, 0, anchor_matching=False, num_organs=4), seg_proxy=False, seg_fg_bg=False)
    assert not fused_criterion.usable(boxes_matching, out, targets, None)
    assert getattr(boxes_matching(out, targets, None, anchors), "vector", None) is None
    one_query = {"pred_logits": logits[-1][:, :4], "pred_boxes": boxes[-1][:, :4]}
    anchor_matching = TransoarCriterion(4, Matcher(1, 0, 0, anchor_matching=True, num_organs=4), seg_proxy=False, seg_fg_bg=False)
    assert not fused_criterion.usable(anchor_matching, one_query, targets, None)
    cpu = {k: v.cpu() for k, v in one_query.items()}
    assert not fused_criterion.usable(anchor_matching, cpu, DenseTargets(tgt.cpu(), present.cpu(), 3), None)
