"""One training step of TransoarNet: autocast forward, weighted loss sum,
backward (with the data-parallel gradient exchange overlapped), AdamW.

The unit the headline metric times.  Follows the step of
transoar/trainer.py:54-92 and the optimiser set-up of scripts/train.py:52-65
(two parameter groups: ``_backbone`` at lr_backbone, the rest at lr; AdamW,
weight decay 1e-4; optional gradient clipping, StepLR(lr_drop) stepped per epoch through end_epoch()), with these deliberate changes:
  * bf16 autocast without a GradScaler (the reference uses fp16 + GradScaler on
    CUDA; bf16 needs no loss scaling) -- fp32 master weights either way;
  * no ``.item()`` in the step (the reference does six host syncs per step,
    trainer.py:87-92): losses are returned as device tensors;
  * targets are densified once per step (DenseTargets) and the loss
    normalisers are all-reduced so N data-parallel replicas equal one process
    on the concatenated batch.
"""
import os

import torch

from .data_parallel import GradientAllReducer
from .matcher import DenseTargets


def build_optimizer(model, config, fused=None, capturable=False):
    """scripts/train.py:52-63.  capturable: the step counters and learning rates live on the device, so that the
    AdamW update can be part of a captured HIP graph (the scheduler then changes the rates in place)."""
    backbone = [p for n, p in model.named_parameters() if "_backbone" in n and p.requires_grad]
    rest = [p for n, p in model.named_parameters() if "_backbone" not in n and p.requires_grad]
    if fused is None:
        fused = all(p.is_cuda for p in backbone + rest)
    lr, lr_backbone = float(config["lr"]), float(config["lr_backbone"])
    if capturable:
        dev = (backbone + rest)[0].device
        lr, lr_backbone = torch.tensor(lr, device=dev), torch.tensor(lr_backbone, device=dev)
    groups = [{"params": backbone}, {"params": rest, "lr": lr}]
    if fused and os.environ.get("TRANSOAR_TORCH_ADAMW", "0") != "1" and all(p.dtype == torch.float32 for p in backbone + rest):
        # the whole update in one launch (transoar_amd/optim.py); torch.optim.AdamW's state and schedulers as they are
        from .optim import FlatAdamW
        return FlatAdamW(groups, lr=lr_backbone, weight_decay=float(config["weight_decay"]))
    return torch.optim.AdamW(groups, lr=lr_backbone, weight_decay=float(config["weight_decay"]), fused=fused, capturable=capturable)


class TrainStep:
    def __init__(self, model, criterion, config, optimizer=None, amp_dtype=torch.bfloat16,
                 process_group=None, bucket_bytes=48 << 20, graph=False, scheduler=None):
        self.model, self.criterion, self.config = model, criterion, config
        # one rank + graph: the AdamW update is captured too (the whole step is one graph launch); with several
        # ranks the gradient exchange sits between backward and update, which stays eager
        single = not (torch.distributed.is_available() and torch.distributed.is_initialized())
        # several ranks + graph (round 4): the bucket all-reduces are captured too -- launched by the gradient hooks from
        # inside the captured backward (RCCL's stream forks off the capture stream: a parallel branch of the graph that
        # overlaps the rest of the backward), joined before the captured AdamW.  TRANSOAR_DP_CAPTURE_EXCHANGE=0: round 3's
        # form (graph = fwd + loss + bwd, then an eager exchange and an eager AdamW: the whole exchange exposed).
        self.capture_exchange = bool(graph and not single and os.environ.get("TRANSOAR_DP_CAPTURE_EXCHANGE", "1") != "0")
        self.capture_optimizer = bool(graph and (single or self.capture_exchange) and optimizer is None
                                      and os.environ.get("TRANSOAR_EAGER_OPTIMIZER") is None and next(model.parameters()).is_cuda)
        self.optimizer = optimizer or build_optimizer(model, config, capturable=self.capture_optimizer)
        # scripts/train.py:65: StepLR(optim, lr_drop), stepped once per EPOCH (trainer.py:220) -> end_epoch()
        self.scheduler = scheduler if scheduler is not None or "lr_drop" not in config else \
            torch.optim.lr_scheduler.StepLR(self.optimizer, int(config["lr_drop"]))
        self.amp_dtype = amp_dtype
        # flat gradient buckets only where gradients are exchanged: on one rank autograd hands every parameter a fresh
        # gradient tensor (no zero-fill of buckets, no per-parameter accumulate kernel: ~170 launches less per step)
        self.reducer = GradientAllReducer(model, process_group, bucket_bytes, always_flat=False)
        self._graph = None
        self._replay_done = None
        self._static_counts = None
        self._static_counts_local = None
        self._expect_total = None
        self._want_graph = graph
        self.num_classes = config["num_classes"]
        self.device_type = next(model.parameters()).device.type

    def _prepack(self):
        """Filter packs of every implicit-GEMM convolution for this step's weights in one launch (conv_gemm.PackPlan)
        instead of two permute + cast copies per layer; part of the captured step."""
        if os.environ.get("TRANSOAR_CONV_PREPACK", "1") == "0":
            return
        plan = getattr(self, "_pack_plan", None)
        if plan is None or not plan.valid():
            from .conv3d import Conv3dK3
            from .conv_gemm import PackPlan
            mods = [m for m in self.model.modules() if isinstance(m, Conv3dK3) and Conv3dK3.enabled and m.uses_gemm()
                    and m.weight.is_cuda and m.weight.dtype == torch.float32 and m.weight.is_contiguous()]
            plan = self._pack_plan = PackPlan(mods) if mods else False
        if plan:
            plan.run()

    def _shadow(self):
        """The bf16 mirrors of the fp32 parameters (transoar_amd/shadow.py), refreshed: one multi-tensor copy per step
        instead of a cast kernel per weight, per bias and per gradient.  None: switched off (TRANSOAR_SHADOW_WEIGHTS=0)."""
        if os.environ.get("TRANSOAR_SHADOW_WEIGHTS", "1") == "0":
            return None
        reg = getattr(self, "_shadow_reg", None)
        if reg is None or not reg.valid():
            from .ms_deform_attn import MSDeformAttn
            from .shadow import ShadowWeights
            stacks = [(m.sampling_offsets.weight, m.attention_weights.weight)
                      for m in self.model.modules() if isinstance(m, MSDeformAttn)]
            reg = self._shadow_reg = ShadowWeights(self.model, stacks)
        reg.refresh()
        return reg

    def end_epoch(self):
        """Advance the learning-rate schedule (the reference steps it after every epoch, trainer.py:220).  The
        fused AdamW reads the group's lr on every step, so this is safe next to a captured graph."""
        if self.scheduler is not None:
            self.scheduler.step()

    def loss(self, data, targets, seg_targets=None, counts=None):
        """-> (weighted total, dict of unweighted losses); forward only.  counts: already
        rank-summed loss normalisers (a 2-element device tensor), else they are reduced here."""
        if not isinstance(targets, DenseTargets):
            targets = DenseTargets.from_list(targets, self.num_classes, data.device)
        if counts is not None:
            targets = DenseTargets(targets.boxes, targets.present, counts[0], counts[1])
        pending = None
        if counts is None and self.reducer.active:
            # the rank-summed normalisers are only needed by the criterion: the sum is launched here and waited for after
            # the model's forward (no blocking collective opens a step)
            counts = torch.stack((torch.as_tensor(float(targets.num_boxes), device=data.device),
                                  targets.present.sum().float()))
            pending = self.reducer.reduce_counts_async(counts)
            targets = DenseTargets(targets.boxes, targets.present, counts[0], counts[1])
        enabled = self.amp_dtype is not None and self.amp_dtype != torch.float32
        mirrors = None
        if enabled and self.amp_dtype == torch.bfloat16 and data.is_cuda and torch.is_grad_enabled():
            self._prepack()
            mirrors = self._shadow()
        from . import shadow
        with shadow.fresh(mirrors), \
                torch.autocast(self.device_type, dtype=self.amp_dtype if enabled else torch.bfloat16, enabled=enabled):
            out = self.model(data)
            if pending is not None:
                pending.wait()
            losses = self.criterion(out, targets, seg_targets, self.model._anchors)
            total = self._weighted_total(losses)
        return total, losses

    def _weighted_total(self, losses):
        """sum_k coef[k] * loss[k] as ONE stack and one dot product (eleven multiplies and ten adds, and as many nodes in the
        backward, were 40 launches of ~5 us per step)."""
        coefs = self.config["loss_coefs"]
        vec = getattr(losses, "vector", None)
        if vec is not None:                      # the fused criterion hands its losses over as one tensor already
            key = (tuple(losses.keys()), vec.device, vec.dtype)
            cached = getattr(self, "_coef_vec", None)
            if cached is None or cached[0] != key:
                w = torch.tensor([float(coefs[k.split("_")[0]]) for k in losses], dtype=vec.dtype, device=vec.device)
                cached = self._coef_vec = (key, w)
            return torch.dot(vec, cached[1])
        vals = list(losses.values())
        dt = vals[0].dtype
        for v in vals[1:]:
            dt = torch.promote_types(dt, v.dtype)
        key = (tuple(losses.keys()), vals[0].device, dt)
        cached = getattr(self, "_coef_vec", None)
        if cached is None or cached[0] != key:
            w = torch.tensor([float(coefs[k.split("_")[0]]) for k in losses], dtype=dt, device=vals[0].device)
            cached = self._coef_vec = (key, w)
        return torch.dot(torch.stack([v.to(dt) for v in vals]), cached[1])

    # ---- captured-graph mode ------------------------------------------------------------------
    # The eager step is close to host-bound on one MI355X (~2000 kernel launches: 56 ms of enqueue for
    # 60 ms of GPU work), so forward + criterion + backward are captured once into a HIP graph over static
    # input buffers and replayed; the gradient exchange (eager, on the flat buckets, after the
    # replay -- no collective inside the graph) and the fused optimizer follow.
    def capture(self, data, targets, warmup=3):
        """Capture fwd+loss+bwd for inputs of this shape.  Raises if anything in the step cannot be
        captured; the caller may then keep using the eager step."""
        assert self._want_graph, "construct TrainStep(graph=True)"
        if not isinstance(targets, DenseTargets):
            targets = DenseTargets.from_list(targets, self.num_classes, data.device)
        self.model.train()
        self._static_x = data.clone()
        self._static_t = DenseTargets(targets.boxes.clone(), targets.present.clone(), targets.num_boxes)
        if getattr(self.criterion, "_seg_proxy", False):
            raise RuntimeError("TrainStep.capture: the segmentation proxy loss needs seg_targets, which the captured "
                               "step does not carry; use the eager step for use_seg_proxy_loss configs")
        # the loss normalisers (number of boxes, number of present classes) live in a static DEVICE buffer that
        # is refilled before every replay -- also on one GPU: a Python int would be baked into the captured
        # graph and mis-scale every batch whose box count differs from the captured one
        self._static_counts = self._local_counts(self._static_t)
        # ... rank-summed.  With the exchange captured the sum is the graph's FIRST node (this rank's counts are written
        # to _static_counts_local before a replay, the graph copies them and all-reduces the copy): no host-launched
        # collective in front of a replay (round-4 VERDICT weak #16).  Otherwise an eager all-reduce before each replay.
        in_graph = self.reducer.active and self.capture_exchange and os.environ.get("TRANSOAR_DP_COUNTS_IN_GRAPH", "1") != "0"
        self._static_counts_local = self._static_counts.clone() if in_graph else None
        if self.reducer.active:
            self.reducer.reduce_counts(self._static_counts)
        self.reducer.overlap = self.capture_exchange       # hooks launch the buckets' all-reduces only when they are captured
        # warm-up AND capture on one and the same side stream: autograd's AccumulateGrad nodes remember the
        # stream of the first backward; if the capture runs on another stream the engine inserts cross-stream
        # waits into the captured graph (the "AccumulateGrad node's stream does not match" warning)
        # capture() must leave the model and the optimizer as it found them: the warm-up steps and the verification
        # replay below update weights and moments when the AdamW update is part of the graph (round-2 ADVICE)
        snap = self._snapshot() if self.capture_optimizer else None
        side = self.capture_stream()
        side.wait_stream(torch.cuda.current_stream())
        eager_total = None
        with torch.cuda.stream(side):
            for k in range(warmup):
                total = self._eager_fwd_bwd(self._static_x, self._static_t)[0]
                if k == 0:
                    eager_total = total          # the loss on the weights capture() was called with
                    if self.reducer.active:
                        self.reducer._end_first_step()      # dead parameters lose their bucket views before the capture (ADVICE)
                if self.capture_exchange:         # warm RCCL's kernels / buffers for these bucket sizes outside the capture
                    self.reducer.exchange()
                if self.capture_optimizer:        # warm the optimizer's kernels (and allocate its state) outside the capture
                    self._clip()
                    self.optimizer.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        from . import graph_replay_safe
        if not graph_replay_safe() and not os.environ.get("TRANSOAR_TRUST_PACKET_CAPTURE"):
            if snap is not None:
                self._restore(snap)
            raise RuntimeError("HIP was initialised with DEBUG_CLR_GRAPH_PACKET_CAPTURE on: this ROCm's pre-recorded "
                               "graph packets corrupt the replayed step (DESIGN.md section 8); export "
                               "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before starting the process, or call transoar_amd.use_safe_graph_replay() before the first HIP call")
        if self.capture_optimizer and hasattr(self.optimizer, "prepare_capture"):
            self.optimizer.prepare_capture()
        if self.reducer.active:
            # The process group's watchdog thread polls hipEventQuery on the end events of the warm-up steps' collectives
            # until it has seen them complete (every 100 ms).  Once the capture pulls the group's internal stream into
            # capture mode, HIP answers such a query with hipErrorCapturedEvent ("event last recorded in a capturing
            # stream" -- it looks at the stream's state now, not at record time), the watchdog rethrows and the process
            # aborts: one capture in four died that way when it started within a poll interval of the last eager
            # collective.  Everything is complete on the GPU (synchronize above); give the watchdog three polls to notice.
            # (ProcessGroupNCCL exposes neither its work queue nor a flush of it to Python, and the poll interval is a
            # compile-time constant of torch: the wait is a multiple of it, TRANSOAR_DP_CAPTURE_SETTLE_MS to change it --
            # e.g. on a host so loaded that the watchdog thread is starved for longer.)
            torch.cuda.synchronize()
            import time
            time.sleep(max(0.0, float(os.environ.get("TRANSOAR_DP_CAPTURE_SETTLE_MS", "300")) * 1e-3))
        graph = torch.cuda.CUDAGraph()
        # with a process group alive its watchdog thread polls events while we capture: only calls made
        # by THIS thread may invalidate the capture
        mode = "thread_local" if self.reducer.active else "global"
        try:
            with torch.cuda.graph(graph, stream=side, capture_error_mode=mode):
                if self._static_counts_local is not None:
                    self._static_counts.copy_(self._static_counts_local)
                    self.reducer.reduce_counts(self._static_counts)
                self._static_total, self._static_losses = self._eager_fwd_bwd(self._static_x, self._static_t)
                if self.capture_exchange:
                    self.reducer.exchange()       # buckets not launched by a hook yet, the joins, the widening copies
                if self.capture_optimizer:
                    self._clip()
                    self.optimizer.step()
        except Exception:
            if snap is not None:
                self._restore(snap)
            raise
        # ---- verification before the captured step is trusted: one replay on the capture batch, on the weights
        # capture() was called with, against the eager loss on those weights (only the dropout masks differ)
        if snap is not None:
            self._restore(snap)
        graph.replay()
        torch.cuda.synchronize()
        got = float(self._static_total)
        expect = float(eager_total) if eager_total is not None else got
        if snap is not None:
            self._restore(snap)          # undo the verification replay's update
        if not (got == got and abs(got - expect) <= 0.2 * abs(expect) + 1e-3):
            self.drop_graph()
            raise RuntimeError("graph replay gives loss %r, the eager step gave %r" % (got, expect))
        self._expect_total = None
        self._graph = graph
        # TRANSOAR_GRAPH_SERIALIZE=1: never launch the graph again before its previous launch has finished on the GPU
        self._replay_done = torch.cuda.Event() if os.environ.get("TRANSOAR_GRAPH_SERIALIZE") else None
        return self

    def _snapshot(self):
        """Copies of every parameter and module buffer (with the version counter each had), of the optimizer's state tensors
        (None where the state does not exist yet) and of the device's RNG state (the warm-up steps and the verification replay
        draw dropout seeds)."""
        params = [(p.detach().clone(), p._version) for p in list(self.model.parameters()) + list(self.model.buffers())]
        self._snap_rng = torch.cuda.get_rng_state(next(self.model.parameters()).device)
        state = []
        for group in self.optimizer.param_groups:
            for p in group["params"]:
                st = self.optimizer.state.get(p)
                state.append(None if not st else {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()})
        return params, state

    def _restore(self, snap):
        """In place (the captured graph holds the addresses): parameters back to the snapshot, optimizer state back to
        the snapshot or -- where it was created after the snapshot -- to its initial zeros.

        A tensor whose version counter still has the snapshot's value was never written by the warm-up steps, and the captured
        step is the same code, so a replay does not write it either: it is left alone.  This is not an optimisation.  copy_
        advances the version counter, and derived data is cached per (tensor, version) -- the RoI key masks on `roi_pad`
        (roi_attn.key_mask), the gathered positional tokens (focused_decoder) -- so a restore that "rewrote" those constant
        buffers made the next eager forward rebuild the derived tensors and FREE the ones whose addresses the graph had just
        baked in: the next tensors allocated on that stream landed in them, and the replay read tile counts out of someone
        else's data (round 6: a memory access fault in roi_attn_fwd)."""
        params, state = snap
        torch.cuda.set_rng_state(self._snap_rng, next(self.model.parameters()).device)
        with torch.no_grad():
            for p, (c, version) in zip(list(self.model.parameters()) + list(self.model.buffers()), params):
                if p._version != version:
                    p.copy_(c)
            i = 0
            for group in self.optimizer.param_groups:
                for p in group["params"]:
                    st, old = self.optimizer.state.get(p), state[i]
                    i += 1
                    if not st:
                        continue
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            if old is None:
                                v.zero_()
                            else:
                                v.copy_(old[k])
        self.model.zero_grad(set_to_none=False)

    def capture_stream(self):
        """The side stream every eager step before a capture should run on (see capture())."""
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream()
        return self._side

    @staticmethod
    def _local_counts(targets):
        return torch.stack((torch.as_tensor(float(targets.num_boxes), device=targets.boxes.device),
                            targets.present.sum().float()))

    def _eager_fwd_bwd(self, data, targets):
        self.reducer.begin()
        total, losses = self.loss(data, targets, counts=getattr(self, "_static_counts", None))
        total.backward()
        return total.detach(), losses

    def _replay(self, data, targets):
        if data.data_ptr() != self._static_x.data_ptr():
            self._static_x.copy_(data)
        if targets is not self._static_t:
            self._static_t.boxes.copy_(targets.boxes)
            self._static_t.present.copy_(targets.present)
            self._static_t.num_boxes = targets.num_boxes
        if self._static_counts_local is not None:
            self._static_counts_local.copy_(self._local_counts(self._static_t))      # summed over the ranks inside the graph
        else:
            self._static_counts.copy_(self._local_counts(self._static_t))
            if self.reducer.active:
                self.reducer.reduce_counts(self._static_counts)
        if self._replay_done is not None:
            self._replay_done.synchronize()      # the previous step (replay + all-reduce + AdamW) has left the GPU: see capture()
        self._graph.replay()
        if self._expect_total is not None:
            expect, self._expect_total = self._expect_total, None
            got = float(self._static_total)
            if not (got == got and abs(got - expect) <= 0.2 * abs(expect) + 1e-3):
                self.drop_graph()
                raise RuntimeError("graph replay gives loss %r, the eager step gave %r" % (got, expect))
        if self.reducer.active and not self.capture_exchange:
            self.reducer.exchange()          # the same path as the eager step's finish(): wire compression included
        if not self.capture_optimizer:
            self._clip()
            self.optimizer.step()
        if self._replay_done is not None:
            self._replay_done.record()
        return self._static_total, self._static_losses

    def drop_graph(self):
        """Back to the eager step (with its overlapped gradient exchange)."""
        self._graph = None
        self._static_counts = None
        self._static_counts_local = None
        self.reducer.overlap = True

    def _clip(self):
        max_norm = self.config.get("clip_max_norm", -1)
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_([p for p in self.model.parameters() if p.grad is not None], max_norm)

    def __call__(self, data, targets, seg_targets=None):
        if self._graph is not None:
            if not isinstance(targets, DenseTargets):
                targets = DenseTargets.from_list(targets, self.num_classes, data.device)
            return self._replay(data, targets)
        self.model.train()
        if self._want_graph and data.is_cuda:
            # an eager step of a TrainStep that will be captured runs on the capture's side stream: autograd's AccumulateGrad
            # nodes (and the gradient hooks that launch RCCL from them) keep the stream of the FIRST backward, and a hook
            # firing on another stream inside the capture ends it with a crash in hipStreamEndCapture (tests/_dp_capture_probe.py)
            side = self.capture_stream()
            if torch.cuda.current_stream(data.device) == side:
                # the caller runs its loop on the capture stream already (`with torch.cuda.stream(step.capture_stream())`): no
                # hand-over.  The two waits below drain the GPU's queue at every step boundary -- 0.9 ms of a 32-ms step on an
                # MI355X (round 6: 32.83 against 31.90 ms) -- so a loop of eager steps of a to-be-captured TrainStep belongs there
                return self._eager_step(data, targets, seg_targets)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                out = self._eager_step(data, targets, seg_targets)
            torch.cuda.current_stream().wait_stream(side)
            return out
        return self._eager_step(data, targets, seg_targets)

    def _eager_step(self, data, targets, seg_targets):
        self.reducer.begin()
        total, losses = self.loss(data, targets, seg_targets)
        total.backward()
        self.reducer.finish()
        self._clip()
        self.optimizer.step()
        return total.detach(), losses
