"""FlatAdamW (transoar_amd/optim.py: every parameter in one launch) against torch.optim.AdamW(fused=True) on the same
gradients: parameters and both moments after several steps, two groups with different learning rates, tensor sizes on
both sides of the chunk size and not divisible by 4, a learning-rate change in between, state_dict round trip, and the
update as part of a captured graph."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(384, 384), (1000,), (7,), (3, 5, 7), (16384,), (16385,), (40000, 3), (1,), (768, 384, 3, 3, 3)]
    return [torch.randn(s, generator=g).cuda().requires_grad_() for s in shapes]


def _pair():
    from transoar_amd.optim import FlatAdamW
    a, b = _params(0), _params(0)
    mine = FlatAdamW([{"params": a[:4]}, {"params": a[4:], "lr": 3e-3}], lr=1e-3, weight_decay=1e-2)
    ref = torch.optim.AdamW([{"params": b[:4]}, {"params": b[4:], "lr": 3e-3}], lr=1e-3, weight_decay=1e-2, fused=True)
    return a, b, mine, ref


def _set_grads(ps, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    for p in ps:
        p.grad = torch.randn(p.shape, device="cuda", generator=g) * (1.0 + seed)


def _close(x, y, tol=2e-6):
    return float((x - y).abs().max()) <= tol * max(float(y.abs().max()), 1e-30)


def test_flat_adamw_matches_torch_fused_adamw():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a, b, mine, ref = _pair()
    for k in range(5):
        _set_grads(a, k)
        _set_grads(b, k)
        if k == 3:                                   # a scheduler's job: change the rates between steps
            mine.param_groups[0]["lr"].fill_(5e-4)
            ref.param_groups[0]["lr"] = 5e-4
        mine.step()
        ref.step()
    torch.cuda.synchronize()
    for p, q in zip(a, b):
        assert _close(p, q), (tuple(p.shape), float((p - q).abs().max()))
        assert _close(mine.state[p]["exp_avg"], ref.state[q]["exp_avg"])
        assert _close(mine.state[p]["exp_avg_sq"], ref.state[q]["exp_avg_sq"])
        assert float(mine.state[p]["step"]) == 5.0
    # a parameter without a gradient is left alone (the Focused Decoder's dead q_proj weights)
    before = a[2].detach().clone()
    _set_grads(a, 9)
    a[2].grad = None
    mine.step()
    assert torch.equal(a[2], before) and float(mine.state[a[2]]["step"]) == 5.0 and float(mine.state[a[0]]["step"]) == 6.0


def test_flat_adamw_state_dict_round_trip_and_version_counters():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from transoar_amd.optim import FlatAdamW
    a, _, mine, _ = _pair()
    _set_grads(a, 0)
    v0 = a[0]._version
    mine.step()
    assert a[0]._version > v0, "the raw-pointer update must bump autograd's version counter"
    import copy
    sd = copy.deepcopy(mine.state_dict())      # (load_state_dict keeps tensors that already have the right dtype and device: a checkpoint on disk is a copy)
    c = _params(0)
    with torch.no_grad():
        for p, q in zip(c, a):
            p.copy_(q)
    other = FlatAdamW([{"params": c[:4]}, {"params": c[4:], "lr": 3e-3}], lr=1e-3, weight_decay=1e-2)
    other.load_state_dict(sd)
    _set_grads(a, 1)
    _set_grads(c, 1)
    mine.step()
    other.step()
    for p, q in zip(a, c):
        assert torch.equal(p, q)


def test_flat_adamw_inside_a_captured_graph():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a, b, mine, ref = _pair()
    static = [torch.zeros_like(p) for p in a]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for p, s in zip(a, static):
            p.grad = s * 2.0                 # a fresh tensor per step, as autograd makes them
        mine.step()                          # warm-up (state allocation) outside the capture
    torch.cuda.current_stream().wait_stream(side)
    for p, q in zip(a, b):                   # bring the torch twin to the same point
        q.grad = torch.zeros_like(q)
    ref.step()
    mine.prepare_capture()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for p, s in zip(a, static):
            p.grad = s * 2.0
        mine.step()
    for k in range(3):
        g = torch.Generator(device="cuda").manual_seed(100 + k)
        for s, q in zip(static, b):
            s.copy_(torch.randn(s.shape, device="cuda", generator=g))
            q.grad = s * 2.0
        graph.replay()
        ref.step()
    torch.cuda.synchronize()
    for p, q in zip(a, b):
        assert _close(p, q), tuple(p.shape)
    with pytest.raises(RuntimeError):        # a second capture without a prepared table
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=side):
            for p, s in zip(a, static):
                p.grad = s * 3.0
            mine.step()
