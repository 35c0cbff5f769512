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
