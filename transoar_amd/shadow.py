"""bf16 shadow copies of the fp32 master weights, refreshed once per training step.

Under bf16 autocast every GEMM / convolution operand that is a parameter is cast fp32 -> bf16 by its own little copy
kernel (one per weight and one per bias, ~4-6 us each), and autograd casts every such gradient back bf16 -> fp32 with
another one: ~230 launches of a 45-ms step whose only content is 54 M parameters = 0.1 ms of HBM time (round-2 VERDICT,
item 6; the reference's counterpart is autocast inside trainer.py:67-77).  Here the step makes ALL casts at once:

  * `ShadowWeights(model)` lays one bf16 mirror of every fp32 parameter into one flat buffer (views at 128-byte
    aligned offsets; parameters named in a `stack` group lie back to back, so that the concatenation of e.g.
    sampling_offsets.weight | attention_weights.weight is a free view) and refreshes it with one multi-tensor copy
    (`torch._foreach_copy_`: a handful of launches for all parameters) -- TrainStep.loss does that next to the
    convolutions' filter packs, inside the captured step, after the previous step's AdamW update;
  * the modules ask `shadow.bf16(param)` where they used to write `param.to(torch.bfloat16)`; outside a training
    step (evaluation, tests on bare modules) the answer is None and they cast as before;
  * `shadow.linear(x, weight, bias)` is F.linear for the small (query-side) GEMMs of the Focused Decoder and the heads:
    bf16 operands from the mirror, and a backward that produces the weight / bias gradients directly in fp32
    (fp32-output GEMM, fp32 column sum) instead of bf16 gradients plus a cast each.

Same arithmetic contract as autocast: operands rounded to bf16 once, fp32 accumulation, bf16 outputs, fp32 parameter
gradients.  (The weight gradient is rounded to bf16 by autocast's path before it is cast back; here it is not.)
"""
import torch
import torch.nn.functional as F

from . import rows

_ALIGN = 64          # elements: 128 bytes of bf16
_current = None      # the registry whose mirrors are fresh: set by `fresh()` around a forward pass


class ShadowWeights:
    def __init__(self, model, stacks=()):
        """stacks: iterable of parameter tuples that must lie back to back (each first dimension-stacked)."""
        params = [p for p in model.parameters() if p.dtype == torch.float32 and p.is_contiguous() and p.numel() > 0]
        if not params or len({p.device for p in params}) != 1:
            raise ValueError("ShadowWeights: needs fp32 parameters on one device")
        self.device = params[0].device
        known = {id(p) for p in params}
        order, placed, self._stacks = [], set(), {}
        for group in stacks:
            group = tuple(group)
            if all(id(p) in known and id(p) not in placed for p in group) and \
                    all(p.numel() % 8 == 0 for p in group[:-1]) and len({tuple(p.shape[1:]) for p in group}) == 1:
                order.append(group)
                placed.update(id(p) for p in group)
        order += [(p,) for p in params if id(p) not in placed]
        total, offsets = 0, []
        for group in order:
            total = -(-total // _ALIGN) * _ALIGN
            offsets.append(total)
            total += sum(p.numel() for p in group)
        self.flat = torch.empty(total, dtype=torch.bfloat16, device=self.device)
        # the refresh writes through an alias of the same storage with its own version counter: a mirror saved for a
        # backward pass stays valid when a second forward refreshes the (unchanged) weights before that backward runs
        alias = torch.empty(0, dtype=torch.bfloat16, device=self.device).set_(self.flat.untyped_storage(), 0, self.flat.shape)
        self.params, self.views, self._writers, self._index = [], [], [], {}
        for group, off in zip(order, offsets):
            start = off
            for p in group:
                v = self.flat[off: off + p.numel()].view(p.shape)
                self.params.append(p)
                self.views.append(v)
                self._writers.append(alias[off: off + p.numel()].view(p.shape))
                self._index[id(p)] = v
                off += p.numel()
            if len(group) > 1:
                rows = sum(p.shape[0] for p in group)
                self._stacks[tuple(id(p) for p in group)] = self.flat[start: off].view(rows, *group[0].shape[1:])
        self._ptrs = [p.data_ptr() for p in self.params]
        # Hard constraint of the mirrors: they are rewritten in place by refresh() through an alias that
        # autograd's version counters do not see, so a backward pass must run BEFORE the refresh that follows a change of
        # the weights (TrainStep's order: forward, backward, optimizer step, refresh).  `generation` counts the refreshes
        # that found changed weights; the autograd Functions that save a mirror stamp it in forward and compare in
        # backward (stamp / check below) -- two graphs kept alive across an optimizer step raise instead of silently
        # differentiating with the new weights.
        self.generation, self._versions = 0, None

    def valid(self):
        """False once a parameter was re-allocated (model.to(...), load with assign=True): rebuild then."""
        return all(p.data_ptr() == q and p.dtype == torch.float32 for p, q in zip(self.params, self._ptrs))

    def refresh(self):
        versions = tuple(p._version for p in self.params)
        if versions != self._versions:
            self.generation += 1
            self._versions = versions
        with torch.no_grad():
            torch._foreach_copy_(self._writers, self.params)

    def get(self, p):
        v = self._index.get(id(p))
        if v is None and getattr(p, "_base", None) is not None and p.is_contiguous():
            base = self._index.get(id(p._base))          # a reshaping view of a parameter (conv.weight.view(co, ci))
            if base is not None and p.numel() == base.numel() and p.storage_offset() == p._base.storage_offset():
                v = base.view(p.shape)
        return v

    def get_stack(self, group):
        return self._stacks.get(tuple(id(p) for p in group))


class fresh:
    """with shadow.fresh(registry): the forward pass inside may read the mirrors (they were refreshed after the last
    parameter update)."""

    def __init__(self, registry):
        self.registry = registry

    def __enter__(self):
        global _current
        self._prev, _current = _current, self.registry
        return self.registry

    def __exit__(self, *exc):
        global _current
        _current = self._prev
        return False


def stamp(ctx):
    """forward of a Function that saves a mirror: remember which state of the weights it saw"""
    ctx._shadow_stamp = None if _current is None else (_current, _current.generation)


def check(ctx):
    """backward of the same Function: the mirrors must still hold the weights of its forward"""
    st = getattr(ctx, "_shadow_stamp", None)
    if st is not None and st[0].generation != st[1]:
        raise RuntimeError("transoar_amd.shadow: the bf16 weight mirrors were refreshed with changed weights between this "
                           "forward and its backward (two graphs kept alive across an optimizer step / weight load): the "
                           "backward would use the new weights.  Run backward before the next step's forward.")


def bf16(p):
    """The fresh bf16 mirror of parameter p, or None (no training step in progress / p not mirrored)."""
    return None if _current is None else _current.get(p)


def bf16_or_cast(p):
    v = bf16(p)
    return p.to(torch.bfloat16) if v is None else v


def bf16_stack(group):
    """The mirror of torch.cat(group) for a registered stack group, or None."""
    return None if _current is None else _current.get_stack(group)


class _Linear(torch.autograd.Function):
    """y = x W^T + b on the bf16 mirrors; fp32 parameter gradients straight from the backward GEMM / column sum."""

    @staticmethod
    def forward(ctx, x, weight, bias, wb, bb):
        xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
        ctx.save_for_backward(xb, wb)
        stamp(ctx)
        ctx.in_dtype, ctx.has_bias = x.dtype, bias is not None
        with torch.autocast("cuda", enabled=False):
            return F.linear(xb, wb, bb)

    @staticmethod
    def backward(ctx, gy):
        check(ctx)
        xb, wb = ctx.saved_tensors
        gx = gw = gb = None
        with torch.autocast("cuda", enabled=False):
            gy2 = gy.reshape(-1, gy.shape[-1])
            if gy2.dtype != torch.bfloat16:
                gy2 = gy2.to(torch.bfloat16)
            if ctx.needs_input_grad[0]:
                gx = (gy2 @ wb).view(xb.shape)
                if gx.dtype != ctx.in_dtype:
                    gx = gx.to(ctx.in_dtype)
            if ctx.needs_input_grad[1]:
                gw = torch.mm(gy2.t(), xb.reshape(-1, xb.shape[-1]), out_dtype=torch.float32)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                gb = rows.colsum_any(gy2)
        return gx, gw, gb, None, None


def linear(x, weight, bias=None):
    """F.linear(x, weight, bias) for a small GEMM under bf16 autocast; the mirror path when the step provides one."""
    wb = bf16(weight)
    if wb is None or not x.is_cuda or not torch.is_autocast_enabled() or x.dtype not in (torch.float32, torch.bfloat16):
        return F.linear(x, weight, bias)
    bb = None if bias is None else bf16(bias)
    if bias is not None and bb is None:
        return F.linear(x, weight, bias)
    return _Linear.apply(x, weight, bias, wb, bb)


class _AsBf16(torch.autograd.Function):
    """param -> its bf16 mirror as a differentiable value (for operands of einsum / matmul expressions): the forward
    cast disappears, the gradient is cast to fp32 as autograd's own `.to()` node would."""

    @staticmethod
    def forward(ctx, param):
        return bf16(param).detach()

    @staticmethod
    def backward(ctx, g):
        return g.float()


def as_bf16(p):
    """p.to(torch.bfloat16) with autograd; reads the mirror inside a training step."""
    m = bf16(p)
    return p.to(torch.bfloat16) if m is None else _AsBf16.apply(p)


def as_dtype(p, dtype):
    """p.to(dtype) with autograd: the mirror for bf16 inside a training step, the parameter itself (no detour through
    bf16: the fp32 evaluation path must not round its weights) for every other dtype."""
    return as_bf16(p) if dtype == torch.bfloat16 else p.to(dtype)


class _SelfAttnProj(torch.autograd.Function):
    """The packed input projection of nn.MultiheadAttention for self-attention with q = k = x_qk and v = x_v:
    (x_qk W[:2C]^T + b[:2C], x_v W[2C:]^T + b[2C:]) from the mirrors; the gradient of the packed weight is assembled
    from two fp32-output GEMMs (autograd's route: three chunk views, six casts and a zero-padded add per chunk)."""

    @staticmethod
    def forward(ctx, x_qk, x_v, weight, bias, wb, bb):
        c = weight.shape[1]
        xq = x_qk if x_qk.dtype == torch.bfloat16 else x_qk.to(torch.bfloat16)
        xv = x_v if x_v.dtype == torch.bfloat16 else x_v.to(torch.bfloat16)
        ctx.save_for_backward(xq, xv, wb)
        stamp(ctx)
        ctx.dtypes = (x_qk.dtype, x_v.dtype)
        with torch.autocast("cuda", enabled=False):
            return F.linear(xq, wb[:2 * c], bb[:2 * c]), F.linear(xv, wb[2 * c:], bb[2 * c:])

    @staticmethod
    def backward(ctx, g_qk, g_v):
        check(ctx)
        xq, xv, wb = ctx.saved_tensors
        c = wb.shape[1]
        with torch.autocast("cuda", enabled=False):
            g1 = g_qk.reshape(-1, 2 * c).to(torch.bfloat16)
            g2 = g_v.reshape(-1, c).to(torch.bfloat16)
            gxq = (g1 @ wb[:2 * c]).view(xq.shape).to(ctx.dtypes[0]) if ctx.needs_input_grad[0] else None
            gxv = (g2 @ wb[2 * c:]).view(xv.shape).to(ctx.dtypes[1]) if ctx.needs_input_grad[1] else None
            gw = torch.cat((torch.mm(g1.t(), xq.reshape(-1, c), out_dtype=torch.float32),
                            torch.mm(g2.t(), xv.reshape(-1, c), out_dtype=torch.float32)))
            gb = torch.cat((rows.colsum_any(g1), rows.colsum_any(g2)))
        return gxq, gxv, gw, gb, None, None


def self_attention(mha, x_qk, x_v):
    """nn.MultiheadAttention(q = k = x_qk, v = x_v, need_weights=False)[0] for batch-first (B, Q, C) inputs on the
    mirrors, or None when the mirror path does not apply (the caller then calls the module)."""
    if not (isinstance(mha, torch.nn.MultiheadAttention) and mha._qkv_same_embed_dim and mha.in_proj_bias is not None
            and mha.bias_k is None and not mha.add_zero_attn and x_qk.is_cuda and torch.is_autocast_enabled()):
        return None
    wb, bb = bf16(mha.in_proj_weight), bf16(mha.in_proj_bias)
    if wb is None or bb is None or bf16(mha.out_proj.weight) is None:
        return None
    b, q, c = x_qk.shape
    h = mha.num_heads
    qk, v = _SelfAttnProj.apply(x_qk, x_v, mha.in_proj_weight, mha.in_proj_bias, wb, bb)
    qk = qk.view(b, q, 2, h, c // h)
    out = F.scaled_dot_product_attention(qk[:, :, 0].transpose(1, 2), qk[:, :, 1].transpose(1, 2),
                                         v.view(b, q, h, c // h).transpose(1, 2),
                                         dropout_p=mha.dropout if mha.training else 0.0)
    return linear(out.transpose(1, 2).reshape(b, q, c), mha.out_proj.weight, mha.out_proj.bias)
