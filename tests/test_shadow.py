"""bf16 weight mirrors (transoar_amd/shadow.py): layout and refresh on the CPU; on the GPU the training step with mirrors
must differentiate the same function as the step with autocast's per-parameter casts."""
import os

import pytest
import torch


def test_mirror_layout_refresh_and_stacks():
    from transoar_amd import shadow
    torch.manual_seed(0)
    net = torch.nn.ModuleDict({"a": torch.nn.Linear(24, 16), "b": torch.nn.Linear(24, 8), "c": torch.nn.Conv3d(8, 4, 1)})
    reg = shadow.ShadowWeights(net, stacks=[(net["a"].weight, net["b"].weight)])
    assert shadow.bf16(net["a"].weight) is None                     # no step in progress
    reg.refresh()
    with shadow.fresh(reg):
        for p in net.parameters():
            m = shadow.bf16(p)
            assert m.dtype == torch.bfloat16 and m.shape == p.shape and m.data_ptr() % 16 == 0
            assert torch.equal(m, p.detach().to(torch.bfloat16))
        st = shadow.bf16_stack((net["a"].weight, net["b"].weight))
        assert st.shape == (24, 24)
        assert torch.equal(st, torch.cat((net["a"].weight, net["b"].weight)).detach().to(torch.bfloat16))
        assert st.data_ptr() == shadow.bf16(net["a"].weight).data_ptr()
        assert shadow.bf16_stack((net["b"].weight, net["a"].weight)) is None
        w2 = net["c"].weight.view(4, 8)                              # a reshaping view of a parameter finds its mirror
        assert torch.equal(shadow.bf16(w2), net["c"].weight.detach().view(4, 8).to(torch.bfloat16))
        saved_version = shadow.bf16(net["a"].weight)._version
        with torch.no_grad():
            net["a"].weight.mul_(2)
        reg.refresh()                                               # through the alias: no version bump on the readers
        assert shadow.bf16(net["a"].weight)._version == saved_version
        assert torch.equal(shadow.bf16(net["a"].weight), net["a"].weight.detach().to(torch.bfloat16))
    assert shadow.bf16(net["a"].weight) is None
    assert reg.valid()
    net["a"].weight.data = net["a"].weight.data.clone()
    assert not reg.valid()


@pytest.mark.gpu
def test_step_with_mirrors_matches_step_with_casts():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from test_train_step_gpu import _batch, _flagship
    from transoar_amd.train_step import TrainStep, build_optimizer
    cfg, model, crit = _flagship()
    step = TrainStep(model, crit, cfg, optimizer=build_optimizer(model, cfg), amp_dtype=torch.bfloat16, graph=False)
    x, t = _batch(cfg, 1)
    params = {n: p for n, p in model.named_parameters() if p.requires_grad}
    grads, losses = {}, {}
    for mode in ("0", "1"):
        os.environ["TRANSOAR_SHADOW_WEIGHTS"] = mode
        try:
            model.zero_grad(set_to_none=True)
            total, _ = step._eager_fwd_bwd(x, t)
        finally:
            os.environ.pop("TRANSOAR_SHADOW_WEIGHTS")
        losses[mode] = float(total)
        grads[mode] = {n: p.grad.detach().double().clone() for n, p in params.items() if p.grad is not None}
    assert step._shadow_reg is not None and step._shadow_reg.valid()
    assert set(grads["0"]) == set(grads["1"])
    assert abs(losses["0"] - losses["1"]) <= 2e-3 * abs(losses["0"]), losses
    worst = sorted(((float((grads["1"][n] - g).norm() / g.norm().clamp_min(1e-30)), n) for n, g in grads["0"].items()),
                   reverse=True)
    print("mirrors vs casts, gradient rel-L2 worst:", worst[:4])
    # same bounds as captured-vs-eager (run-to-run atomics order behind the InstanceNorm chain); the weight gradients of
    # the small linears are no longer rounded to bf16 on their way to fp32 (<= 2^-9 relative per element)
    assert worst[0][0] <= 4e-2, worst[:5]
    rest = [w for w in worst if not w[1].startswith("_backbone._encoder.")]
    assert rest[0][0] <= 1.5e-2, rest[:5]


@pytest.mark.gpu
def test_shared_token_gradient_matches_autograd_chain():
    """Focused Decoder cross-attention with keys = values + sine positions: the single token gradient of _FoldedCore
    (dS^T qf + P^T dctx accumulated in one buffer, handed to the values) against autograd's separate key / value chains."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from test_train_step_gpu import _batch, _flagship
    from transoar_amd.focused_decoder import FocusedAttn
    from transoar_amd.train_step import TrainStep, build_optimizer
    cfg, model, crit = _flagship()
    step = TrainStep(model, crit, cfg, optimizer=build_optimizer(model, cfg), amp_dtype=torch.bfloat16, graph=False)
    x, t = _batch(cfg, 2)
    params = {n: p for n, p in model.named_parameters() if p.requires_grad}
    grads, losses = {}, {}
    assert FocusedAttn.shared_token_grad
    from transoar_amd import roi_attn
    fused = roi_attn.ENABLED
    roi_attn.ENABLED = False          # _FoldedCore is the function under test (the fused kernel of round 4 replaces it in the model
    try:                              # and keeps the scores in fp32 where both torch chains round them to bf16: tests/test_roi_attn_gpu.py)
        for mode in (False, True):
            FocusedAttn.shared_token_grad = mode
            model.zero_grad(set_to_none=True)
            total, _ = step._eager_fwd_bwd(x, t)
            losses[mode] = float(total)
            grads[mode] = {n: p.grad.detach().double().clone() for n, p in params.items() if p.grad is not None}
    finally:
        FocusedAttn.shared_token_grad = True
        roi_attn.ENABLED = fused
    assert set(grads[False]) == set(grads[True])
    assert abs(losses[False] - losses[True]) <= 1e-3 * abs(losses[False]), losses        # same forward arithmetic
    worst = sorted(((float((grads[True][n] - g).norm() / g.norm().clamp_min(1e-30)), n) for n, g in grads[False].items()),
                   reverse=True)
    print("shared token gradient vs autograd chain, gradient rel-L2 worst:", worst[:4])
    assert worst[0][0] <= 4e-2, worst[:5]
    rest = [w for w in worst if not w[1].startswith("_backbone._encoder.")]
    assert rest[0][0] <= 1.5e-2, rest[:5]


@pytest.mark.parametrize("follow", [False, True])
def test_folded_attention_keeps_fp32_weights_outside_autocast(follow):
    """The folded per-organ attention against explicit K / V projections in fp32 (no autocast, no mirrors): the weights
    must enter unrounded (a detour through bf16 here cost the fp32 GPU parity of golden g7 a factor 10)."""
    from transoar_amd.focused_decoder import FocusedAttn
    torch.manual_seed(3)
    b, n_org, qpo, n_keys, c, h = 2, 3, 4, 10, 32, 4
    attn = FocusedAttn(c, h, torch.zeros(n_org * qpo, 1), qkv_bias=True).eval()
    q = torch.randn(b, n_org * qpo, c)
    v_tok = torch.randn(b, n_org * n_keys, c)
    k_tok = v_tok + 0.3 * torch.randn(1, n_org * n_keys, c)
    pad = torch.zeros(n_org, n_keys, dtype=torch.bool)
    pad[1, 7:] = True
    got = attn._roi_attention_folded(q, k_tok, v_tok, pad, n_org, n_keys, follow)
    hd = c // h
    kk = attn.k_proj(k_tok).view(b, n_org, n_keys, h, hd).permute(0, 1, 3, 2, 4)
    vv = attn.v_proj(v_tok).view(b, n_org, n_keys, h, hd).permute(0, 1, 3, 2, 4)
    qq = (attn.k_proj(q) * attn.scale).view(b, n_org, qpo, h, hd).permute(0, 1, 3, 2, 4)
    p = (qq @ kk.transpose(-2, -1)).masked_fill(pad[None, :, None, None, :], float("-inf")).softmax(-1)
    want = (p @ vv).permute(0, 1, 3, 2, 4).reshape(b, n_org * qpo, c)
    assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item()


@pytest.mark.gpu
def test_focused_decoder_add_norm_on_the_fused_kernel():
    """FocusedDecoderLayer._add_norm: norm(x + branch) of the decoder sub-layers on csrc/tokens.hip against the eager
    chain under bf16 autocast (fp32 stream + bf16 branch), values and gradients."""
    from transoar_amd.focused_decoder import FocusedDecoderLayer
    torch.manual_seed(5)
    dev = "cuda"
    norm = torch.nn.LayerNorm(384).to(dev)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5); norm.bias.uniform_(-0.3, 0.3)
    drop = torch.nn.Dropout(0.1).eval()
    x0 = torch.randn(2, 540, 384, device=dev)
    r0 = torch.randn(2, 540, 384, device=dev).to(torch.bfloat16)
    g32 = torch.randn(2, 540, 384, device=dev)
    g16 = torch.randn(2, 540, 384, device=dev).to(torch.bfloat16)
    res = {}
    for fused in (True, False):
        FocusedDecoderLayer.fused_norms = fused
        try:
            x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
            norm.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y32, y16 = FocusedDecoderLayer._add_norm(None, x.expand(2, 540, 384), r, norm, drop)
                assert (y16 is not None) == fused and y32.dtype == torch.float32
                y16 = y32.to(torch.bfloat16) if y16 is None else y16
            ((y32 * g32).sum() + (y16.float() * g16.float()).sum()).backward()
            res[fused] = (y32.detach(), y16.detach().float(), x.grad, r.grad.float(), norm.weight.grad.clone(),
                          norm.bias.grad.clone())
        finally:
            FocusedDecoderLayer.fused_norms = True
    for name, a, b in zip(("y32", "y16", "gx", "gr", "gw", "gb"), res[True], res[False]):
        tol = 2e-2 if name in ("gr", "y16") else 2e-4         # rounded to bf16 on both routes: one ulp apart at most
        assert (a - b).abs().max().item() <= tol * b.abs().max().item(), name


def test_mirror_generation_guards_a_backward_that_comes_too_late():
    """The mirrors are rewritten in place behind autograd's back: a Function that saved a mirror stamps the
    registry's generation in forward and refuses in backward when a refresh has meanwhile seen changed weights."""
    from transoar_amd import shadow
    net = torch.nn.Linear(8, 8)
    reg = shadow.ShadowWeights(net, stacks=[])
    reg.refresh()
    g0 = reg.generation
    reg.refresh()                                   # nothing changed: same generation (an evaluation pass between two steps)
    assert reg.generation == g0

    class Ctx:
        pass
    ctx = Ctx()
    with shadow.fresh(reg):
        shadow.stamp(ctx)
    shadow.check(ctx)                               # backward right after: fine
    with torch.no_grad():
        net.weight.add_(1.0)                        # an optimizer step
    reg.refresh()
    assert reg.generation == g0 + 1
    with pytest.raises(RuntimeError, match="mirrors were refreshed"):
        shadow.check(ctx)
    free = Ctx()
    shadow.stamp(free)                              # no registry in force: nothing to guard
    shadow.check(free)
