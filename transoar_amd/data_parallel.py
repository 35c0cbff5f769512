"""Data-parallel gradient exchange for the training step: one process per GPU,
``torch.distributed`` ("nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests).

The reference is single-process (scripts/train.py:28 pins one device; no
collective anywhere), so this has no counterpart to mirror; the
contract it implements is "N replicas on batch shards == one process on the
concatenated batch":

  * the loss normalisers (number of target boxes / valid class slots,
    criterion.py:96) are summed over ranks first (one 2-element all-reduce),
  * gradients are SUMMED over ranks (not averaged).

Mechanics, chosen for xGMI (7 point-to-point links per GPU: a few large
messages, launched early):
  * parameters are packed, in reverse registration order (~ the order autograd
    finishes them: heads -> neck -> FPN decoder -> encoder stage 5 ... 0), into
    a few flat fp32 buckets (default 48 MiB; the model has ~217 MB of grads).
    Autograd hands every parameter a FRESH gradient tensor; when the last gradient
    of a bucket has landed, ONE multi-tensor copy packs the bucket and the
    parameters' ``.grad`` become views of it (what the optimizer then reads).
    (Round 3 made the views the ``.grad`` up front: autograd then accumulates into
    them with one read-modify-write kernel per parameter -- ~170 launches and a
    217-MB zero fill per step, most of the 0.9 ms a one-rank group cost);
  * a post-accumulate-grad hook counts a bucket's parameters down and launches
    its all-reduce asynchronously the moment the last one lands, so the
    exchange of the big early buckets (encoder stage 5: 96 MB) hides behind the
    expensive full-resolution backbone backward that is still to run;
  * parameters that never receive a gradient (the dead ``cross_attn.q_proj``,
    SURVEY F8) are discovered in the first step and excluded afterwards;
  * optional wire compression (``compress="bf16"`` / TRANSOAR_DP_COMPRESS=bf16):
    a bucket travels as a bf16 copy (half the bytes per xGMI link: 108 instead
    of 217 MB per step) and is widened back into the fp32 bucket after the
    wait; the sum over ranks is then a sum of bf16-rounded gradients (2^-9
    relative per element and rank).  Off by default: the step is not
    exchange-bound on the estimates of DESIGN.md section 6.
"""
import os

import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params, device, dtype, wire_dtype=None):
        self.params = params
        # every view starts on a 128-byte boundary of the bucket: torch's multi-tensor kernels (the pack copy below, fused
        # AdamW reading .grad) take their 16-byte vector path only when EVERY tensor of a launch is 16-byte aligned -- with
        # views packed back to back AdamW on bucket views cost 0.71 ms per step against 0.45 on separate gradients
        # (profiles/r04_bench_one_rank_rccl*.json).  The padding elements stay zero: they are summed and ignored.
        align = max(1, 128 // torch.empty((), dtype=dtype).element_size())
        total = sum(-(-p.numel() // align) * align for p in params)
        self.flat = torch.zeros(total, device=device, dtype=dtype)
        # what the collective moves: the bucket itself, or its rounded copy
        self.wire = None if wire_dtype in (None, dtype) else torch.zeros(total, device=device, dtype=wire_dtype)
        self.views, off = [], 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += -(-p.numel() // align) * align
        self.expected = len(params)
        self.pending = self.expected
        self.handle = None


class GradientAllReducer:
    def __init__(self, module, process_group=None, bucket_bytes=48 << 20, always_flat=False, compress=None):
        """always_flat: build the flat gradient buckets even for one rank (the captured-graph
        training step needs gradients at fixed addresses that can be zeroed with a few memsets).
        compress: None (fp32 on the wire) or "bf16"; default from TRANSOAR_DP_COMPRESS."""
        if compress is None:
            compress = os.environ.get("TRANSOAR_DP_COMPRESS") or None
        if compress not in (None, "bf16"):
            raise ValueError("GradientAllReducer: compress must be None or 'bf16', got %r" % (compress,))
        self.wire_dtype = torch.bfloat16 if compress == "bf16" else None
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (dist.is_initialized() and bool(os.environ.get("TRANSOAR_FORCE_DP")))
        self.flat = self.active or always_flat
        self.overlap = True          # launch a bucket's all-reduce from the autograd hook
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.buckets, self._bucket_of, self._hooks = [], {}, []
        self._next = 0               # first bucket of this step that has not been launched yet
        self._fired, self._first_step, self._dead = set(), True, set()
        if not self.flat:
            return
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._add_bucket(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._add_bucket(cur)
        # the hooks also find, in the first step, the parameters that never receive a gradient (one rank too:
        # with flat buckets their .grad would otherwise be a zero view and AdamW would decay them, unlike the
        # reference where their .grad stays None)
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _add_bucket(self, params):
        b = _Bucket(list(params), params[0].device, params[0].dtype, self.wire_dtype if self.active else None)
        for p in b.params:
            self._bucket_of[p] = b
        self.buckets.append(b)

    # ---- per step -----------------------------------------------------------
    def reduce_counts(self, counts):
        """Sum a small tensor of loss normalisers over the ranks (no-op for one rank)."""
        if self.active:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)
        return counts

    def reduce_counts_async(self, counts):
        """The same sum, launched without waiting: returns a handle whose wait() must precede the first use of `counts`
        (None for one rank).  The eager step opens with it and waits only in front of the criterion, so that no blocking
        collective stands between two steps (on RCCL the wait is a stream dependency, not a host wait)."""
        if self.active:
            return dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return None

    def begin(self):
        """Call before forward/backward: drop the old gradients (autograd allocates fresh ones; a bucket is packed when
        its last gradient has landed)."""
        for p in self.params:
            p.grad = None
        if not self.flat:
            return
        for b in self.buckets:
            b.pending = b.expected
            b.handle = None
        self._next = 0
        self._fired.clear()

    def _pack(self, b):
        """The bucket's fresh gradients -> its flat buffer in one multi-tensor copy; .grad becomes the view."""
        live = [(p, v) for p, v in zip(b.params, b.views) if p.grad is not None]
        if live:
            torch._foreach_copy_([v for _, v in live], [p.grad for p, _ in live])
            for p, v in live:
                p.grad = v
        # a parameter of the bucket without a gradient this step contributes ZERO to the exchange, not last step's slice
        # (first step: every parameter that has not fired; later: a live parameter whose branch was skipped on this rank).
        # Its .grad becomes the view all the same: after the exchange it holds the other ranks' sum, and every rank must
        # step the same parameter set or the replicas drift apart (a parameter that no rank ever trains is reset to None by
        # _end_first_step)
        for p, v in zip(b.params, b.views):
            if p.grad is None and p not in self._dead:
                v.zero_()
                p.grad = v

    def _on_grad(self, p):
        b = self._bucket_of[p]
        if p in self._dead:
            # the bucket no longer counts it (b.expected): packing would start before the live gradients are in, and the
            # other ranks would not expect the slice
            raise RuntimeError("GradientAllReducer: a parameter that had no gradient in the first step received one later; "
                               "rebuild the reducer when the set of trained parameters changes")
        if self._first_step:
            self._fired.add(p)
        b.pending -= 1
        if b.pending == 0:
            self._pack(b)
            if self.overlap and self.active:
                self._launch_ready()

    def _launch_ready(self):
        """Launch, in BUCKET ORDER, every packed bucket up to the first one that is not complete yet.  Every rank must issue
        its collectives in the same order; launched the moment each bucket fills up, a rank on which some bucket stays
        incomplete until the end of the backward (a live parameter whose branch it skipped this step) would issue that bucket
        last while the other ranks issue it in the middle -- mismatched collectives (gloo aborts, RCCL hangs or sums the wrong
        buffers; tests/test_data_parallel.py::test_live_parameter_without_gradient...).  Buckets are numbered in the order
        autograd completes them, so in the normal case this launches exactly when the hook fires."""
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if b.pending != 0 or b.handle is not None:
                break
            b.handle = self._launch(b)
            self._next += 1

    def _launch(self, b):
        if b.wire is not None:
            b.wire.copy_(b.flat)
        return dist.all_reduce(b.flat if b.wire is None else b.wire, op=dist.ReduceOp.SUM, group=self.group,
                               async_op=True)

    def finish(self):
        """Call after backward, before the optimizer: wait for every bucket."""
        if not self.active:
            self._end_first_step()
            return
        if not self.overlap:         # captured-graph step: the exchange runs after the replay
            for b in self.buckets:
                b.handle = None
        self.exchange()
        self._end_first_step()

    def exchange(self):
        """Launch every bucket that is not in flight yet, wait for all of them and widen compressed buckets back
        (the tail of the eager step, and the whole exchange of the captured step, which has no hooks running)."""
        for b in self.buckets:       # in bucket order, like the hooks (see _launch_ready)
            if b.handle is None:     # a parameter without a gradient held the bucket -- and every later one -- back
                if b.pending > 0:
                    self._pack(b)
                    b.pending = 0
                b.handle = self._launch(b)
        self._next = len(self.buckets)
        for b in self.buckets:
            b.handle.wait()
            if b.wire is not None:
                b.flat.copy_(b.wire)
            b.handle = None

    def _end_first_step(self):
        if not (self._first_step and self.flat):
            return
        for b in self.buckets:
            b.expected = sum(1 for p in b.params if p in self._fired)
            if b.pending > 0:                      # (one rank, no exchange: a bucket held back by a dead parameter)
                self._pack(b)
                b.pending = 0
            for p in b.params:
                if p not in self._fired:          # dead parameter: no gradient, the optimizer skips it
                    self._dead.add(p)
                    p.grad = None
        self._first_step = False

    def remove(self):
        for h in self._hooks:
            h.remove()
