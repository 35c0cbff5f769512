"""Host-side mirror (modules, refine block, backbone, whole model, criterion)
against golden vectors generated from the reference (tests/golden/make_golden.py).

CPU variants run the package's PyTorch code with the oracle's torch core
injected for ``use_cuda=False`` (the package itself has no CPU compute path);
GPU variants (``-m gpu``) run the same checks with ``use_cuda=True`` through
the gfx950 kernels.  Tolerance: fp32 model, 1e-4 relative to each tensor's max
magnitude (north_star); fp64 fixtures 1e-9.
"""
import os

import numpy as np
import pytest
from tests._observe import observe
import torch

from oracle.torch_ref import msda3d_core_torch
from tests._inputs import (analytic_volume, fill_deterministic, level_starts, small_backbone_config,
                           small_model_config)


def relerr(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-300)


@pytest.fixture
def debug_core():
    from transoar_amd import ms_deform_attn as mod
    prev = mod.register_debug_core(msda3d_core_torch)
    yield
    mod.register_debug_core(prev)


def _device_params():
    return [pytest.param("cpu", id="cpu-oracle-core"),
            pytest.param("cuda", id="gpu-hip", marks=pytest.mark.gpu)]


def _skip_if_no_gpu(device):
    if device == "cuda" and not torch.cuda.is_available():
        pytest.skip("no GPU")


def _load_state(module, z, prefix="state."):
    sd = {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}
    module.load_state_dict(sd, strict=True)


@pytest.mark.parametrize("device", _device_params())
def test_g4_msdeformattn_module(golden_dir, debug_core, device):
    _skip_if_no_gpu(device)
    from transoar_amd import MSDeformAttn
    z = np.load(os.path.join(golden_dir, "g4_module.npz"))
    mod = MSDeformAttn(48, 2, 6, 4, use_cuda=(device == "cuda")).double()
    _load_state(mod, z)
    mod = mod.to(device)
    t = lambda k: torch.from_numpy(z[k]).to(device)
    shapes = t("shapes")
    y = mod(t("query"), t("ref"), t("src"), shapes, level_starts(shapes))
    assert relerr(y, z["y"]) <= 1e-9
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad(y, list(params.values()), t("gy"))
    for name, g in zip(params, grads):
        assert relerr(g, z["grad." + name]) <= 1e-8, name


def test_msdeformattn_init_matches_reference_scheme():
    """Offsets start as k voxels along the head's axis, attention uniform
    (ms_deform_attn.py:63-91); the g4 state was re-randomised, so check init here."""
    from transoar_amd import MSDeformAttn
    m = MSDeformAttn(48, 3, 6, 4)
    b = m.sampling_offsets.bias.view(6, 3, 4, 3)
    dirs = torch.tensor([(-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 1), (0, 1, 0), (1, 0, 0)], dtype=torch.float32)
    for p in range(4):
        assert torch.equal(b[:, :, p], (dirs * (p + 1))[:, None, :].expand(6, 3, 3))
    assert float(m.sampling_offsets.weight.abs().max()) == 0
    assert float(m.attention_weights.weight.abs().max()) == 0 and float(m.attention_weights.bias.abs().max()) == 0
    with pytest.raises(ValueError):
        MSDeformAttn(48, 3, 8, 4)
    assert MSDeformAttn(52, 1, 26, 1).sampling_offsets.bias.view(26, 3).abs().sum(1).min() >= 1


@pytest.mark.parametrize("device", _device_params())
def test_g5_refine_block_on_32cube_pyramid(golden_dir, debug_core, device):
    """BASELINE.json config #1 as restated in SURVEY F5."""
    _skip_if_no_gpu(device)
    from transoar_amd.refine_block import DecoderDefAttnBlock
    z = np.load(os.path.join(golden_dir, "g5_refine_block.npz"))
    blk = DecoderDefAttnBlock(d_model=48, nhead=6, num_layers=2, dim_feedforward=64, dropout=0.1,
                              feature_levels=["P3", "P4", "P5"], n_points=4,
                              use_cuda=(device == "cuda")).double().eval()
    _load_state(blk, z)
    blk = blk.to(device)
    fmaps = [torch.from_numpy(z["fmap%d" % i]).to(device) for i in range(3)]
    pos = [torch.from_numpy(z["pos%d" % i]).to(device) for i in range(3)]
    outs = blk(fmaps, pos)
    for i, o in enumerate(outs):
        assert tuple(o.shape) == tuple(z["out%d" % i].shape)
        assert relerr(o, z["out%d" % i]) <= 1e-9


def test_g5_position_encoding_matches(golden_dir):
    from transoar_amd.position_encoding import PositionEmbeddingSine3D
    z = np.load(os.path.join(golden_dir, "g5_refine_block.npz"))
    enc = PositionEmbeddingSine3D(channels=48)
    for i in range(3):
        f = torch.from_numpy(z["fmap%d" % i])
        assert relerr(enc(f), z["pos%d" % i]) <= 1e-6


@pytest.mark.parametrize("device", _device_params())
@pytest.mark.parametrize("tag,refine", [("plain", False), ("refine", True)])
def test_g6_backbone(golden_dir, debug_core, device, tag, refine):
    _skip_if_no_gpu(device)
    from transoar_amd.backbone import AttnFPN
    z = np.load(os.path.join(golden_dir, "g6_backbone.npz"))
    net = AttnFPN(small_backbone_config(refine, use_cuda=(device == "cuda"))).eval()
    fill_deterministic(net)
    net = net.to(device)
    out = net(analytic_volume((32, 32, 64)).to(device))
    names = [k.split(".", 1)[1] for k in z.files if k.startswith(tag + ".P")]
    assert sorted(out.keys()) == sorted(names)
    for k in names:
        assert relerr(out[k], z["%s.%s" % (tag, k)]) <= 1e-4, k
    total = sum(o.sum() for o in out.values())
    params = dict(net.named_parameters())
    assert list(params.keys()) == list(z[tag + ".grad_names"])
    grads = torch.autograd.grad(total, list(params.values()), allow_unused=True)
    for name, g, s, a in zip(params, grads, z[tag + ".grad_sums"], z[tag + ".grad_abs_sums"]):
        got = 0.0 if g is None else g.double().sum().item()
        # checksums of fp32 gradients through 12 InstanceNorm layers are ill-conditioned (see
        # tests/test_data_parallel.py): CPU-vs-CPU they move with the thread count torch's convolution blocks for
        # (1.2e-4 at 8 threads, 3.2e-4 at 4 -- whatever an earlier test of the session left set): 6e-4; 5e-3 CPU golden vs GPU kernels
        tol = 6e-4 if device == "cpu" else 5e-3
        assert abs(got - s) <= tol * max(a, 1e-6) + 1e-7, (name, got, float(s), float(a))


@pytest.mark.parametrize("device", _device_params())
@pytest.mark.parametrize("tag,refine", [("plain", False), ("refine", True)])
def test_g7_whole_model_and_criterion(golden_dir, debug_core, device, tag, refine):
    _skip_if_no_gpu(device)
    from transoar_amd.config import synthetic_targets
    from transoar_amd.transoarnet import TransoarNet, build_criterion
    z = np.load(os.path.join(golden_dir, "g7_whole_model.npz"))
    cfg = small_model_config(refine, use_cuda=(device == "cuda"))
    net = TransoarNet(cfg).eval()
    fill_deterministic(net)
    net = net.to(device)
    assert relerr(net._anchors, z[tag + ".anchors"]) <= 1e-6
    assert relerr(net._restrictions, z[tag + ".restrictions"]) <= 1e-6
    assert np.array_equal(net._neck.decoder.layers[0].attn_mask.sum(1).cpu().numpy(), z[tag + ".attn_mask_rowsum"])
    out = net(analytic_volume((160, 160, 256), batch=2).to(device))
    assert relerr(out["pred_logits"], z[tag + ".pred_logits"]) <= 1e-4
    assert relerr(out["pred_boxes"], z[tag + ".pred_boxes"]) <= 1e-4
    for i, aux in enumerate(out["aux_outputs"]):
        assert relerr(aux["pred_logits"], z["%s.aux%d_logits" % (tag, i)]) <= 1e-4
        assert relerr(aux["pred_boxes"], z["%s.aux%d_boxes" % (tag, i)]) <= 1e-4
    crit = build_criterion(cfg)
    losses = crit(out, synthetic_targets(2, 20, seed=1, device=device), None, net._anchors)
    assert list(losses.keys()) == list(z[tag + ".loss_names"])
    for (k, v), ref in zip(losses.items(), z[tag + ".loss_values"]):
        assert abs(float(v) - ref) <= 1e-4 * max(abs(ref), 1e-3), k
    coefs = cfg["loss_coefs"]
    total = sum(v * coefs[k.split("_")[0]] for k, v in losses.items())
    assert abs(float(total) - float(z[tag + ".total"])) <= 1e-4 * abs(float(z[tag + ".total"]))
    params = dict(net.named_parameters())
    assert list(params.keys()) == list(z[tag + ".grad_names"])
    grads = torch.autograd.grad(total, list(params.values()), allow_unused=True)
    # the dead q_proj weights are exactly the params without a gradient
    assert [g is None for g in grads] == list(z[tag + ".grad_is_none"])
    assert sorted(n for n, g in zip(params, grads) if g is None) == sorted(
        "_neck.decoder.layers.%d.cross_attn.q_proj.weight" % i for i in range(3))
    bad = []
    for name, g, s, a in zip(params, grads, z[tag + ".grad_sums"], z[tag + ".grad_abs_sums"]):
        if g is None:
            continue
        tol = 1e-3 if device == "cpu" else 5e-2
        if device != "cpu" and name.startswith("_backbone._encoder._stages.0."):
            # the very first convs sit behind 12 InstanceNorms: their fp32 gradient checksum moves by
            # >10 % between two CPU evaluations already (tests/test_data_parallel.py); observed on the GPU 0.124 (refine) /
            # 0.038 (plain), everything else 0.026 against its 0.05 (profiles/r05_observed_errors.json): 2x observed
            tol = 0.25
        if device != "cpu":
            key = "g7.%s.grad_checksum.%s" % (tag, "encoder_stage0" if name.startswith("_backbone._encoder._stages.0.") else "rest")
            observe(key, abs(g.double().sum().item() - s) / max(a, 1e-6), tol)
        if abs(g.double().sum().item() - s) > tol * max(a, 1e-6) + 1e-7:
            bad.append((name, g.double().sum().item(), s, a))
    assert not bad, bad[:5]


@pytest.mark.gpu
@pytest.mark.parametrize("refine", [False, True])
def test_bf16_training_path_reaches_every_parameter(debug_core, refine):
    """The bf16-autocast TRAINING path on the GPU (hand-written conv / InstanceNorm / layout / token
    kernels, which the fp32 golden tests above do not take) must back-propagate into every parameter
    the fp32 path reaches and stay within the stated bf16 tolerances of outputs, losses and gradients.  (A raw kernel call in a tracked
    forward once cut the graph between FPN decoder and encoder without failing any parity test.)"""
    _skip_if_no_gpu("cuda")
    from transoar_amd.config import synthetic_targets
    from transoar_amd.conv3d import Conv3dK3
    from transoar_amd.transoarnet import TransoarNet, build_criterion
    cfg = small_model_config(refine, use_cuda=True)
    # stock random initialisation (the closed-form weights of the golden fixtures make every query alike and the
    # encoder gradients vanish: a pathological operating point for a bf16 comparison); the heads start at zero
    # in the reference (no gradient would reach the body): un-zero them
    torch.manual_seed(0)
    net = TransoarNet(cfg)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() > 1 and float(p.abs().max()) == 0:
                torch.nn.init.xavier_uniform_(p)
    net = net.cuda().train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    x = torch.rand(1, 1, 160, 160, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    targets = synthetic_targets(1, 20, seed=1, device="cuda")
    crit = build_criterion(cfg)
    coefs = cfg["loss_coefs"]
    grads, outs, loss_vals = {}, {}, {}
    # the criterion's top-1 matching flips between near-tied queries under bf16 noise (its bbox loss moves by 5 %
    # while the boxes agree to 6e-4): that changes the function, not the numerics -> the bf16 pass reuses the
    # matches of the fp32 pass, so both passes differentiate the same function
    recorded, assign = [], crit.matcher.assign
    old_min = Conv3dK3.min_voxels
    try:
        for mode in ("fp32", "bf16"):
            Conv3dK3.min_voxels = 0 if mode == "bf16" else old_min      # take the hand-written conv path too
            if mode == "fp32":
                crit.matcher.assign = lambda *a, **k: recorded.append(assign(*a, **k)) or recorded[-1]
            else:
                replay = iter(recorded)
                crit.matcher.assign = lambda *a, **k: next(replay)
            net.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(mode == "bf16")):
                out = net(x)
                losses = crit(out, targets, None, net._anchors)
            total = sum(v * coefs[k.split("_")[0]] for k, v in losses.items())
            total.backward()
            outs[mode] = {k: out[k].detach().float() for k in ("pred_logits", "pred_boxes")}
            loss_vals[mode] = {k: float(v) for k, v in losses.items()}
            grads[mode] = {n: (None if p.grad is None else p.grad.detach().double().flatten().cpu())
                           for n, p in net.named_parameters()}
    finally:
        Conv3dK3.min_voxels = old_min
        crit.matcher.assign = assign
    missing = [n for n, g in grads["bf16"].items() if g is None and grads["fp32"][n] is not None]
    assert not missing, missing
    # ---- stated bf16 tolerances of the model, against the fp32 run of the same kernels' host model on the
    # same weights (which the g7 test pins to the reference at 1e-4).  bf16 carries 8 mantissa bits and every
    # activation of the 12-conv backbone + 2 refine layers + 3 decoder layers is rounded to it:
    #   pred_boxes (values in [0,1])        max abs error   <= 5e-3      (observed 1.4e-3)
    #   pred_logits                          max error       <= 0.1 of the largest |logit| + 2e-2 (observed 0.053; rms 0.013)
    #   each of the 11 loss scalars          relative error  <= 1e-2      (observed 2e-3, same matches in both passes)
    #   parameter gradients, relative L2     median <= 6e-2 (observed 0.03), 90 % <= 0.4 (observed 0.26), each <= 0.6 except two named
    #                                        (observed 1.14 on the first InstanceNorm's bias)
    # (the first encoder convolutions sit behind 12 InstanceNorms: their gradients are ill-conditioned --
    # >10 % checksum drift between two fp32 CPU runs, tests/test_data_parallel.py -- hence the tail bound)
    o32, o16 = outs["fp32"], outs["bf16"]
    assert float((o16["pred_boxes"].float() - o32["pred_boxes"]).abs().max()) <= 5e-3
    lmax = float(o32["pred_logits"].abs().max())
    assert float((o16["pred_logits"].float() - o32["pred_logits"]).abs().max()) <= 0.1 * lmax + 2e-2
    for k, v32 in loss_vals["fp32"].items():
        assert abs(loss_vals["bf16"][k] - v32) <= 1e-2 * abs(v32) + 1e-3, (k, loss_vals["bf16"][k], v32)
    rel = []
    for n, g32 in grads["fp32"].items():
        g16 = grads["bf16"][n]
        if g32 is None or float(g32.norm()) < 1e-9:
            continue
        rel.append((float((g16 - g32).norm() / g32.norm()), n))
    rel.sort()
    print("bf16 vs fp32 gradient rel-L2: median %.3g, p90 %.3g, max %.3g (%s)" % (
        rel[len(rel) // 2][0], rel[int(0.9 * len(rel))][0], rel[-1][0], rel[-1][1]))
    observe("model.bf16_vs_fp32.grad_rel_l2.median", rel[len(rel) // 2][0], 6e-2)
    observe("model.bf16_vs_fp32.grad_rel_l2.p90", rel[int(0.9 * len(rel))][0], 0.4)
    observe("model.bf16_vs_fp32.grad_rel_l2.max_outside_first_norm", max(r for r, n in rel if "_stages.0._block.1." not in n), 0.6)
    observe("model.bf16_vs_fp32.grad_rel_l2.first_norm", max([r for r, n in rel if "_stages.0._block.1." in n] or [0.0]), 2.2)
    assert rel[len(rel) // 2][0] <= 6e-2, rel[len(rel) // 2]
    assert rel[int(0.9 * len(rel))][0] <= 0.4, rel[int(0.9 * len(rel)):][:5]
    # per-tensor bound (round-2 VERDICT weak #2): every tensor <= 0.6 (observed <= 0.39) except the two named ones -- the
    # affine parameters of the FIRST InstanceNorm, behind all 12 norm layers of the encoder, where two fp32 runs already
    # differ by > 10 % (tests/test_data_parallel.py): observed 0.58 / 1.14
    # (round 5, profiles/r05_observed_errors.json: median 0.032, p90 0.265, max elsewhere 0.387, the two named ones 1.09)
    ill = {"_backbone._encoder._stages.0._block.1.weight": 2.2, "_backbone._encoder._stages.0._block.1.bias": 2.2}
    over = [(r, n) for r, n in rel if r > ill.get(n, 0.6)]
    assert not over, over


@pytest.mark.parametrize("device", _device_params())
@pytest.mark.parametrize("tag,conv_merging", [("linear_merge", False), ("conv_merge", True)])
def test_g8_swin_encoder_backbone(golden_dir, debug_core, device, tag, conv_merging):
    """SURVEY 8 row f-3 / BASELINE config #4: AttnFPN with use_encoder_attn=True.  Swin stages 2-5 (window and
    shifted-window attention with padding, a grid with one window along two axes, a single-window stage),
    both patch-merge variants: encoder maps, FPN outputs, state_dict names (checkpoint keys) and the parameter
    gradients of sum(outputs) against the reference's."""
    _skip_if_no_gpu(device)
    from tests._inputs import small_swin_config
    from transoar_amd.backbone import AttnFPN
    z = np.load(os.path.join(golden_dir, "g8_swin_backbone.npz"))
    net = AttnFPN(small_swin_config(conv_merging)).eval()
    assert list(net.state_dict().keys()) == list(z[tag + ".state_names"])
    fill_deterministic(net)
    net = net.to(device)
    x = analytic_volume((32, 32, 64)).to(device)
    enc = net._encoder(x)
    # the conv patch merge ends in an InstanceNorm over 2x2x4 / 1x1x2 voxels at C4 / C5: normalising over 2 values
    # amplifies the fp32 differences between CPU and GPU kernels (2e-4 observed at C5 on the GPU)
    tol = 1e-4 if device == "cpu" or not conv_merging else 1e-3
    for k in ("C2", "C3", "C4", "C5"):
        assert relerr(enc[k], z["%s.%s" % (tag, k)]) <= tol, k
    out = net(x)
    for k, v in out.items():
        assert relerr(v, z["%s.%s" % (tag, k)]) <= tol, k
    total = sum(o.sum() for o in out.values())
    params = dict(net.named_parameters())
    assert list(params.keys()) == list(z[tag + ".grad_names"])
    grads = torch.autograd.grad(total, list(params.values()), allow_unused=True)
    gtol = 2e-4 if device == "cpu" else (5e-3 if not conv_merging else 5e-2)
    for name, g, s, a in zip(params, grads, z[tag + ".grad_sums"], z[tag + ".grad_abs_sums"]):
        got = 0.0 if g is None else g.double().sum().item()
        # (a weight directly in front of a norm layer has an analytically zero gradient: fp32 noise ~1e-7)
        assert abs(got - s) <= gtol * max(a, 1e-6) + 2e-6, name


def _g9_stage(golden_dir):
    from transoar_amd.swin_encoder import EncoderSwinBlock
    z = np.load(os.path.join(golden_dir, "g9_swin_stage.npz"))
    stage = EncoderSwinBlock(dim=96, depth=2, num_heads=3, window_size=(5, 5, 5), mlp_ratio=4, qkv_bias=True, qk_scale=None,
                             drop=0.0, attn_drop=0.0, drop_path=0.0, downsample=None).eval()
    assert list(stage.state_dict().keys()) == list(z["state_names"])
    fill_deterministic(stage)
    with torch.no_grad():
        for blk in stage.blocks:
            t = blk.attn.relative_position_bias_table
            t.copy_(0.3 * torch.sin(torch.arange(t.numel(), dtype=torch.float32).view_as(t) * 0.37))
    return z, stage


def test_g9_full_width_swin_stage_fp32(golden_dir):
    """One Swin stage at stage 2's flagship width (96 channels, 3 heads of 32, two blocks: plain + shifted windows) on a
    7x6x11 grid against the reference's EncoderSwinBlock (tests/golden/make_golden.py:g9_swin_stage): the plain torch
    formulation in fp32, CPU."""
    z, stage = _g9_stage(golden_dir)
    x = torch.from_numpy(z["x"]).requires_grad_()
    y = stage(x)
    assert relerr(y, z["y"]) <= 1e-5
    params = dict(stage.named_parameters())
    grads = torch.autograd.grad((y * torch.from_numpy(z["g"])).sum(), [x] + list(params.values()))
    assert relerr(grads[0], z["dx"]) <= 1e-4
    for name, g in zip(params, grads[1:]):
        assert relerr(g, z["grad." + name]) <= 1e-4, name


@pytest.mark.gpu
def test_g9_full_width_swin_stage_on_the_kernels(golden_dir, monkeypatch):
    """The same stage under bf16 autocast on the GPU: window partition / merge on the row kernels, the projections and
    the MLP on the hand-written GEMMs, attention on the window-attention kernel (checked: it is called once per block each
    way) -- against the reference's fp32 numbers at bf16 tolerances."""
    _skip_if_no_gpu("cuda")
    from transoar_amd import swin_encoder, win_attn
    z, stage = _g9_stage(golden_dir)
    stage = stage.cuda()
    calls = []
    real = win_attn.window_attention
    monkeypatch.setattr(swin_encoder, "MIN_TOKENS", 0)          # 462 tokens: take the GEMM kernels anyway
    monkeypatch.setattr(win_attn, "window_attention", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    x = torch.from_numpy(z["x"]).cuda().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = stage(x)
    assert len(calls) == 2, calls
    assert relerr(y, z["y"]) <= 2.0 ** -6, relerr(y, z["y"])
    params = dict(stage.named_parameters())
    grads = torch.autograd.grad((y.float() * torch.from_numpy(z["g"]).cuda()).sum(), [x] + list(params.values()))
    assert relerr(grads[0], z["dx"]) <= 2.0 ** -5, relerr(grads[0], z["dx"])
    for name, g in zip(params, grads[1:]):
        # sums over the 462 tokens of bf16-rounded terms; the 96-element norm / bias vectors are the noisiest (observed 0.04)
        assert relerr(g, z["grad." + name]) <= (8e-2 if g.dim() == 1 else 3e-2), (name, relerr(g, z["grad." + name]))


def test_swin_stochastic_depth_and_window_layout():
    """DropPath drops whole samples with probability p and rescales the survivors; the one-gather window
    layout equals pad + roll + partition written out."""
    from transoar_amd.swin_encoder import DropPath, effective_window, window_layout
    torch.manual_seed(0)
    dp = DropPath(0.25).train()
    y = dp(torch.ones(4000, 3, 2))
    kept = y[:, 0, 0] != 0
    assert torch.all(y[kept] == 1 / 0.75) and abs(float(kept.float().mean()) - 0.75) < 0.03
    assert torch.equal(dp.eval()(torch.ones(5, 2)), torch.ones(5, 2))
    grid, window, shift = (4, 7, 12), (5, 5, 5), (2, 2, 2)
    win, sh = effective_window(grid, window, shift)
    assert win == (4, 5, 5) and sh == (0, 2, 2)
    lay = window_layout(grid, win, sh, "cpu")
    x = torch.arange(4 * 7 * 12, dtype=torch.float32).view(1, 4, 7, 12, 1) + 1
    pad = torch.nn.functional.pad(x, (0, 0, 0, 3, 0, 3, 0, 0))                      # -> (4, 10, 15)
    rolled = torch.roll(pad, shifts=(0, -2, -2), dims=(1, 2, 3))
    ref = rolled.view(1, 1, 4, 2, 5, 3, 5, 1).permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(1, 6, 100, 1)
    tok = torch.cat((x.view(1, -1, 1), torch.zeros(1, 1, 1)), 1)
    got = tok[:, lay.gather].view(1, lay.n_windows, lay.n_per, 1)
    assert torch.equal(got, ref)
    assert torch.equal(got.reshape(1, -1, 1)[:, lay.scatter], x.view(1, -1, 1))
