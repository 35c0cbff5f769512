"""AMOS geometry (BASELINE.json config #1's whole-model half): 256x256x128 volume, 3-level pyramid
(32,32,16),(16,16,8),(8,8,4), 15 organs x 27 queries = 405 queries, neck on P3
(config/attn_fpn_foc_dec_amos.yaml).  Whole-model eval forward and one bf16 training step on the GPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(refine=True):
    from transoar_amd.config import amos_config, synthetic_bbox_properties
    from transoar_amd.transoarnet import TransoarNet, build_criterion
    cfg = amos_config(refine=refine, use_cuda=True)
    cfg["bbox_properties"] = synthetic_bbox_properties(cfg["num_classes"], seed=0)
    torch.manual_seed(0)
    return cfg, TransoarNet(cfg).cuda(), build_criterion(cfg)


def test_amos_whole_model_eval_forward():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg, net, _ = _model()
    net.eval()
    assert tuple(net._neck.decoder.layers[0].attn_mask.shape) == (405, 32 * 32 * 16)
    x = torch.rand(1, 1, 256, 256, 128, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    with torch.no_grad():
        out32 = net(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out16 = net(x)
    for out in (out32, out16):
        assert tuple(out["pred_logits"].shape) == (1, 405, 1) and tuple(out["pred_boxes"].shape) == (1, 405, 6)
        assert torch.isfinite(out["pred_logits"].float()).all() and torch.isfinite(out["pred_boxes"].float()).all()
        assert float(out["pred_boxes"].min()) >= 0 and float(out["pred_boxes"].max()) <= 1
    # bf16 autocast against fp32 on the same weights: stated model tolerance (tests/test_model_parity.py)
    assert float((out16["pred_boxes"].float() - out32["pred_boxes"]).abs().max()) <= 1e-2
    lmax = float(out32["pred_logits"].abs().max())
    assert float((out16["pred_logits"].float() - out32["pred_logits"]).abs().max()) <= 3e-2 * lmax + 1e-2


def test_amos_training_step_reaches_every_live_parameter():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from transoar_amd.config import synthetic_targets
    from transoar_amd.matcher import DenseTargets
    from transoar_amd.train_step import TrainStep
    cfg, net, crit = _model()
    step = TrainStep(net, crit, cfg, amp_dtype=torch.bfloat16, graph=False)
    x = torch.rand(1, 1, *cfg["volume_shape"], device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    targets = DenseTargets.from_list(synthetic_targets(1, cfg["num_classes"], seed=1, device="cuda"), cfg["num_classes"], "cuda")
    losses = []
    for _ in range(3):
        total, _ = step(x, targets)
        losses.append(float(total))
    assert all(l == l and abs(l) < 1e4 for l in losses), losses
    none = [n for n, p in net.named_parameters() if p.grad is None]
    assert none and all("cross_attn.q_proj" in n for n in none), none      # the only dead parameters
    bad = [n for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    assert not bad, bad[:5]
