"""Fused residual-add + LayerNorm (+ casts, + positional query) against the stock torch chain."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _stock(x, r, norm, pos_sine, level_embed, level_of_token):
    s = x.float() + (0 if r is None else r.float())
    y32 = torch.nn.functional.layer_norm(s, (x.shape[-1],), norm.weight.float(), norm.bias.float(), norm.eps)
    y16 = y32.to(torch.bfloat16)
    q16 = None
    if pos_sine is not None:
        q16 = (y32 + (pos_sine + level_embed[level_of_token])).to(torch.bfloat16)
    return y32, y16, q16


@pytest.mark.parametrize("cols", [128, 384, 1024])
@pytest.mark.parametrize("x_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_r,with_q", [(True, True), (True, False), (False, True)])
def test_add_layernorm_forward_backward(cols, x_dtype, with_r, with_q):
    from transoar_amd import tokens
    torch.manual_seed(0)
    n, sizes = 2, [700, 90, 13]
    s_tok = sum(sizes)
    dev = "cuda"
    norm = torch.nn.LayerNorm(cols).to(dev)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5); norm.bias.uniform_(-0.5, 0.5)
    x0 = (torch.randn(n, s_tok, cols, device=dev) * 2 + 0.3).to(x_dtype)
    r0 = torch.randn(n, s_tok, cols, device=dev).to(torch.bfloat16) if with_r else None
    pos_sine = torch.randn(s_tok, cols, device=dev) if with_q else None
    le0 = torch.randn(len(sizes), cols, device=dev) if with_q else None
    lot = torch.repeat_interleave(torch.arange(len(sizes), device=dev), torch.as_tensor(sizes, device=dev))
    starts = torch.tensor([0, sizes[0], sizes[0] + sizes[1]], dtype=torch.int32, device=dev)
    g32 = torch.randn(n, s_tok, cols, device=dev)
    g16 = torch.randn(n, s_tok, cols, device=dev).to(torch.bfloat16)
    gq = torch.randn(n, s_tok, cols, device=dev).to(torch.bfloat16)

    def run(fused):
        x = x0.clone().requires_grad_(True)
        r = None if r0 is None else r0.clone().requires_grad_(True)
        le = None if le0 is None else le0.clone().requires_grad_(True)
        norm.zero_grad()
        if fused:
            y32, y16, q16 = tokens.add_layernorm(x, r, norm, *((pos_sine, le, starts) if with_q else ()))
        else:
            y32, y16, q16 = _stock(x, r, norm, pos_sine, le, lot)
        loss = (y32 * g32).sum() + (y16.float() * g16.float()).sum()
        if with_q:
            loss = loss + (q16.float() * gq.float()).sum()
        loss.backward()
        return (y32, y16, q16, x.grad, None if r is None else r.grad, norm.weight.grad.clone(), norm.bias.grad.clone(),
                None if le is None else le.grad)

    a, b = run(True), run(False)
    names = ["y32", "y16", "q16", "gx", "gr", "gw", "gb", "g_level_embed"]
    for name, u, v in zip(names, a, b):
        if v is None:
            assert u is None or name == "q16", name
            continue
        assert u.dtype == v.dtype, name
        u, v = u.double(), v.double()
        scale = v.abs().max().clamp_min(1e-30)
        tol = 2.0 ** -7 if a[names.index(name)].dtype == torch.bfloat16 else 2e-5
        if name in ("gw", "gb", "g_level_embed"):
            tol = 1e-4          # long column sums, different summation order
        assert float((u - v).abs().max() / scale) <= tol, (name, float((u - v).abs().max() / scale))


def test_rejects_bad_width():
    from transoar_amd import tokens
    assert not tokens.usable(torch.zeros(2, 4, 100, device="cuda"), None, 100)


def test_dropout_inside_add_layernorm_and_relu_dropout():
    """With an active nn.Dropout the kernels apply the byte mask themselves: same result as the
    stock chain fed the same mask, forward and backward."""
    from transoar_amd import tokens
    torch.manual_seed(1)
    dev, cols, rows = "cuda", 384, 1000
    norm = torch.nn.LayerNorm(cols).to(dev)
    x0 = torch.randn(1, rows, cols, device=dev)
    r0 = torch.randn(1, rows, cols, device=dev).to(torch.bfloat16)
    keep = tokens.dropout_mask(r0, 0.1)
    frac = keep.float().mean().item()
    assert 0.88 < frac < 0.92
    scale = 1.0 / 0.9
    g = torch.randn(1, rows, cols, device=dev)
    res = []
    for fused in (True, False):
        x = x0.clone().requires_grad_(True); r = r0.clone().requires_grad_(True)
        norm.zero_grad()
        if fused:
            y32, y16, _ = tokens._AddLayerNorm.apply(x, r, norm.weight, norm.bias, norm.eps, None, None, None, keep, scale)
        else:
            y32 = torch.nn.functional.layer_norm(x + (r.float() * keep * scale), (cols,), norm.weight, norm.bias, norm.eps)
        (y32 * g).sum().backward()
        res.append((y32, x.grad, r.grad))
    for u, v in zip(*res):
        tol = 2.0 ** -7 if u.dtype == torch.bfloat16 else 2e-5
        assert u.dtype == v.dtype
        assert float((u.double() - v.double()).abs().max() / v.double().abs().max()) <= tol
    # relu + dropout
    h0 = torch.randn(rows, 1024, device=dev).to(torch.bfloat16)
    keep2 = tokens.dropout_mask(h0, 0.1)
    gy = torch.randn(rows, 1024, device=dev).to(torch.bfloat16)
    h = h0.clone().requires_grad_(True)
    y = tokens._ReluDropout.apply(h, keep2, scale)
    y.backward(gy)
    h2 = h0.clone().requires_grad_(True)
    y2 = (torch.relu(h2).float() * keep2 * scale).to(torch.bfloat16)
    y2.backward(gy)
    assert float((y.float() - y2.float()).abs().max()) <= 2.0 ** -7 * float(y2.float().abs().max())
    assert float((h.grad.float() - h2.grad.float()).abs().max()) <= 2.0 ** -7 * float(h2.grad.float().abs().max())
    # inactive dropout: exact relu
    drop = torch.nn.Dropout(0.1).eval()
    assert torch.equal(tokens.relu_dropout(h0, drop), torch.relu(h0))



@pytest.mark.parametrize("geom", [(2, 6, 4, 4, ((5, 5, 8), (3, 3, 4), (2, 2, 2), (1, 1, 2))), (1, 6, 3, 4, ((4, 4, 4), (2, 2, 2), (1, 1, 1))),
                                  (2, 8, 2, 2, ((3, 4, 5), (2, 2, 3)))])
@pytest.mark.parametrize("shared_ref", [True, False])
def test_sampling_head_matches_eager_chain(geom, shared_ref):
    """The fused MSDeformAttn head (offsets -> locations, logits -> softmax) against the eager autocast chain of
    ms_deform_attn.py:114-128: same rounding points, so forward agrees to fp32 rounding of the softmax and the
    gradient of the projection to one bf16 ulp of the largest entry."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.nn.functional as F
    from transoar_amd import tokens
    n, m, lv, pt, shapes = geom
    torch.manual_seed(n * 100 + m)
    lq = sum(d * h * w for d, h, w in shapes)
    spatial = torch.as_tensor(shapes, dtype=torch.long, device="cuda")
    proj = (torch.randn(n, lq, 4 * m * lv * pt, device="cuda") * 1.5).to(torch.bfloat16).requires_grad_()
    ref = torch.rand(1 if shared_ref else n, lq, lv, 3, device="cuda")
    assert tokens.sampling_head_usable(proj, ref, spatial, m, lv, pt)
    loc, attn = tokens.sampling_head(proj, ref, spatial, m, lv, pt)
    g_loc, g_attn = torch.randn_like(loc), torch.randn_like(attn)
    (g_proj,) = torch.autograd.grad([loc, attn], [proj], [g_loc, g_attn])

    p2 = proj.detach().clone().requires_grad_()
    n_off = m * lv * pt * 3
    with torch.autocast("cuda", dtype=torch.bfloat16):
        offsets = p2[..., :n_off].unflatten(-1, (m, lv, pt, 3))
        w = F.softmax(p2[..., n_off:].unflatten(-1, (m, lv * pt)), dim=-1).view(n, lq, m, lv, pt)
        whd = spatial.flip(-1).to(offsets.dtype)
        loc_ref = ref[:, :, None, :, None, :] + offsets / whd[None, None, None, :, None, :]
    assert loc_ref.dtype == torch.float32 and w.dtype == torch.float32
    (g_ref,) = torch.autograd.grad([loc_ref, w], [p2], [g_loc, g_attn])
    assert torch.equal(loc, loc_ref.expand_as(loc))
    assert float((attn - w).abs().max()) <= 1e-6
    err = float((g_proj.float() - g_ref.float()).abs().max())
    assert err <= 2.0 ** -7 * float(g_ref.float().abs().max()), err


def test_seeded_dropout_matches_hashed_mask():
    """Dropout from a per-call seed (no mask tensor): the kernels' mask equals tokens.hashed_keep() of the same
    seed, forward and backward, and keeps the requested fraction."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transoar_amd import tokens
    torch.manual_seed(3)
    dev, cols, rows, p = "cuda", 384, 1200, 0.1
    scale = 1.0 / (1.0 - p)
    seed = tokens.dropout_seed(torch.zeros(1, device=dev))
    assert seed.dtype == torch.int32 and seed.numel() == 1
    keep = tokens.hashed_keep(seed, rows * cols, 1.0 - p).view(1, rows, cols)
    assert 0.89 < keep.float().mean().item() < 0.91
    assert not torch.equal(keep, tokens.hashed_keep(tokens.dropout_seed(seed), rows * cols, 1.0 - p).view(1, rows, cols))
    norm = torch.nn.LayerNorm(cols).to(dev)
    x0 = torch.randn(1, rows, cols, device=dev)
    r0 = torch.randn(1, rows, cols, device=dev).to(torch.bfloat16)
    g = torch.randn(1, rows, cols, device=dev)
    res = []
    for k in (seed, keep):
        x = x0.clone().requires_grad_(True); r = r0.clone().requires_grad_(True)
        norm.zero_grad()
        y32, y16, _ = tokens._AddLayerNorm.apply(x, r, norm.weight, norm.bias, norm.eps, None, None, None, k, scale)
        (y32 * g).sum().backward()
        res.append((y32, y16, x.grad, r.grad, norm.weight.grad.clone()))
    for u, v in zip(*res):
        assert torch.equal(u, v)
    h0 = torch.randn(rows, 1024, device=dev).to(torch.bfloat16)
    keep2 = tokens.hashed_keep(seed, h0.numel(), 1.0 - p).view_as(h0)
    assert torch.equal(tokens._ReluDropout.apply(h0, seed, scale), tokens._ReluDropout.apply(h0, keep2, scale))


def test_dropout_seed_changes_on_every_graph_replay():
    """The per-call dropout seed is drawn by torch's generator INSIDE the captured step: every replay must see a
    new one (a seed frozen at capture time would repeat the same dropout mask every step)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transoar_amd import tokens
    like = torch.zeros(1, device="cuda")
    out = torch.zeros(1, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out.copy_(tokens.dropout_seed(like))        # warm-up on the side stream
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out.copy_(tokens.dropout_seed(like))
    seen = set()
    for _ in range(4):
        g.replay()
        torch.cuda.synchronize()
        seen.add(int(out.item()))
    assert len(seen) == 4, seen


@pytest.mark.parametrize("cols", [128, 384, 1024])
def test_pos_query_matches_the_eager_chain(cols):
    """First-layer query of the refine block: bit-equal to round(tokens + (sine + level embedding)); the tokens'
    gradient is the query's, level_embed's the per-level column sums."""
    from transoar_amd import tokens
    torch.manual_seed(1)
    n, sizes, dev = 2, [1100, 170, 23, 5], "cuda"
    s_tok = sum(sizes)
    x0 = torch.randn(n, s_tok, cols, device=dev).to(torch.bfloat16)
    pos_sine = torch.randn(s_tok, cols, device=dev)
    le0 = torch.randn(len(sizes), cols, device=dev)
    starts = torch.tensor([sum(sizes[:i]) for i in range(len(sizes))], dtype=torch.int32, device=dev)
    gq = torch.randn(n, s_tok, cols, device=dev).to(torch.bfloat16)

    def run(fused):
        x = x0.clone().requires_grad_(True)
        le = le0.clone().requires_grad_(True)
        if fused:
            q = tokens.pos_query(x, pos_sine, le, starts)
        else:
            level_pos = torch.cat([le[l].expand(n_l, -1) for l, n_l in enumerate(sizes)], 0)
            q = (x + (pos_sine + level_pos)).to(torch.bfloat16)
        q.backward(gq)
        return q, x.grad, le.grad

    (q, gx, gle), (q_ref, gx_ref, gle_ref) = run(True), run(False)
    assert q.dtype == torch.bfloat16 and torch.equal(q, q_ref)
    assert torch.equal(gx, gx_ref)
    assert (gle - gle_ref).abs().max().item() <= 1e-4 * gle_ref.abs().max().item()


@pytest.mark.parametrize("cols", [48, 96, 192, 384, 512, 8, 768, 1536, 1000])
@pytest.mark.parametrize("xdt", [torch.bfloat16, torch.float32])
def test_layernorm_rows_matches_torch(cols, xdt):
    """Short-row LayerNorm (Swin stages: 48 .. 384 channels) against F.layer_norm in fp32: output, input gradient and the
    weight / bias gradients (per-wave partial sums + one column-sum)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from transoar_amd import tokens
    g = torch.Generator(device="cuda").manual_seed(cols)
    n_rows = 70001
    x = (torch.randn(n_rows, cols, device="cuda", generator=g) * 2 + 0.5).to(xdt).requires_grad_()
    norm = torch.nn.LayerNorm(cols).cuda()
    with torch.no_grad():
        norm.weight.copy_(1 + 0.3 * torch.randn(cols, device="cuda", generator=g))
        norm.bias.copy_(0.2 * torch.randn(cols, device="cuda", generator=g))
    assert tokens.layernorm_rows_usable(x, norm)
    y = tokens.layernorm_rows(x, norm)
    gy = torch.randn(n_rows, cols, device="cuda", generator=g).to(torch.bfloat16)
    y.backward(gy)
    xr = x.detach().float().requires_grad_()
    wr, br = norm.weight.detach().clone().requires_grad_(), norm.bias.detach().clone().requires_grad_()
    yr = torch.nn.functional.layer_norm(xr, (cols,), wr, br, norm.eps)
    yr.backward(gy.float())
    rel = lambda a, b: float((a.float() - b.float()).abs().max()) / max(float(b.float().abs().max()), 1e-30)
    assert y.dtype == torch.bfloat16 and rel(y, yr) <= 2.0 ** -8
    assert x.grad.dtype == xdt and rel(x.grad, xr.grad) <= (2.0 ** -7 if xdt == torch.bfloat16 else 1e-4)
    assert rel(norm.weight.grad, wr.grad) <= 1e-3 and rel(norm.bias.grad, br.grad) <= 1e-3


@pytest.mark.parametrize("xdt", [torch.bfloat16, torch.float32])
def test_layernorm_rows_fork_sums_the_shortcut_gradient_in_its_backward(xdt):
    """(x, LayerNorm(x)) as one node (tokens.layernorm_rows_fork, the Swin block's pre-norm shortcut): the same values, and the
    input gradient = shortcut gradient + the norm's, summed inside the backward kernel (fp32, one rounding) instead of by an
    accumulation pass; a missing shortcut / branch gradient is handled."""
    from transoar_amd import tokens
    g = torch.Generator(device="cuda").manual_seed(3)
    n_rows, cols = 50003, 96
    x0 = (torch.randn(n_rows, cols, device="cuda", generator=g) * 2 + 0.5).to(xdt)
    norm = torch.nn.LayerNorm(cols).cuda()
    gy = torch.randn(n_rows, cols, device="cuda", generator=g).to(torch.bfloat16)
    gs = torch.randn(n_rows, cols, device="cuda", generator=g).to(xdt)
    x = x0.clone().requires_grad_()
    xs, y = tokens.layernorm_rows_fork(x, norm)
    assert torch.equal(xs, x0) and xs.data_ptr() == x.data_ptr()
    torch.autograd.backward((xs, y), (gs, gy))
    fused, gw, gb = x.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone()
    x = x0.clone().requires_grad_()
    norm.zero_grad()
    y2 = tokens.layernorm_rows(x, norm)
    assert torch.equal(y, y2)
    y2.backward(gy)
    want = x.grad.float() + gs.float()
    tol = 2.0 ** -7 if xdt == torch.bfloat16 else 1e-6
    assert (fused.float() - want).abs().max().item() <= tol * want.abs().max().item()
    assert torch.equal(gw, norm.weight.grad) and torch.equal(gb, norm.bias.grad)
    # only the shortcut is used / only the branch is used
    x = x0.clone().requires_grad_()
    xs, y = tokens.layernorm_rows_fork(x, norm)
    xs.backward(gs)
    assert torch.equal(x.grad, gs)
    x = x0.clone().requires_grad_()
    xs, y = tokens.layernorm_rows_fork(x, norm)
    y.backward(gy)
    x2 = x0.clone().requires_grad_()
    tokens.layernorm_rows(x2, norm).backward(gy)
    assert torch.equal(x.grad, x2.grad)
