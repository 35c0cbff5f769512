#!/usr/bin/env python
"""Headline benchmark: training-step CT volumes/s of the Focused-Decoder model
on synthetic 160x160x256 volumes (BASELINE.json metric), 1..8 MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = forward + criterion + backward + gradient all-reduce + AdamW on one
batch of 2 volumes per GPU (batch_size of config/attn_fpn_foc_dec_visceral.yaml),
VISCERAL geometry, deformable-attention refinement ON
(backbone.use_decoder_attn=True, use_cuda=True -> the gfx950 MSDeformAttn
kernels; both are off in the shipped yaml), bf16 autocast, fp32
master weights, random-init weights, synthetic data resident in HBM.

Prints ONE JSON line (rank 0).  Besides the driver's contract fields it carries
  roofline      the MSDeformAttn kernel with the largest share of the timed
                region: algorithmic bytes / its average launch
                duration measured with HIP events on the launch stream
  msda_kernels  the same for every MSDeformAttn kernel
  msda_backward the backward chain of the operator against SURVEY's B_bwd
  step_ms       median / p10 / p90 of the per-step hipEvent times
  cpu_baseline  bounded sample timed in this run: ONE whole use_cuda=False training step (fwd + criterion +
                bwd + AdamW, one volume, oracle torch core; kind "port") on the host cores, in volumes/s;
                the operator alone beside it in seconds per call
  cpu_step      the 3-iteration protocol of the same step, measured separately
                (--cpu-baseline-only --cpu-baseline-step, minutes) and quoted from profiles/
"""
import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # see transoar_amd/__init__.py; before torch loads HIP
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC for RCCL between ranks; must be set before HSA initialises

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Round 3: no convolution of the default model goes through MIOpen any more (csrc/conv3d.hip, csrc/conv_gemm.hip), so
# the bench no longer points MIOpen at a tuned find-db.  The one left is the k = s = 2 patch-merging convolution of the
# Swin encoder (BASELINE config #4, --swin): for that run only, tools/miopen_db/ (tuned once on an MI355X) is used.
_DB = os.path.join(ROOT, "tools", "miopen_db")
if "--swin" in sys.argv and os.path.isdir(_DB) and "MIOPEN_USER_DB_PATH" not in os.environ:
    os.environ["MIOPEN_USER_DB_PATH"] = _DB

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def msda_algorithmic_bytes(kind, N, S, M, C, L, Lq, P, e, e_loc):
    """bytes one call must move if every tensor is touched once.  Only the forward gather and the
    backward as a whole have an algorithmic figure; the backward's internal kernels (point sort, walks) move
    workspace bytes of this build's own making, which are not roofline credit."""
    value, out = e * N * S * M * C, e * N * Lq * M * C
    loc_attn = e_loc * N * Lq * M * L * P * 4
    return {
        "fwd": value + out + loc_attn,                                   # B_fwd
        "bwd": value + out + 2 * value + 2 * loc_attn,                   # B_bwd: value, grad_out in; grad_value zero +
    }.get(kind, 0)                                                       # accumulate; loc/attn in, their grads out


BWD_KINDS = ("bwd_query", "cell_count", "scan", "cell_fill", "pull", "value_tile", "value_cells", "bwd_generic")
PMC_FILE = "r06_msda_pmc_step.json"
# the backward as the training step runs it (round 4: transoar_msda3d_backward_proj, bf16 grad_proj instead of fp32
# grad_loc / grad_attn): FETCH_SIZE / WRITE_SIZE passes of `tools/bench_msda.py --proj`
PMC_BWD_FILE = "r06_msda_pmc_step.json"


def pmc_traffic(kind, dims):
    """HBM bytes per launch from the committed PMC collection (profiles/<PMC_FILE>:
    FETCH_SIZE/WRITE_SIZE, separate rocprofv3 --pmc passes, gfx950 correction applied), valid only
    for the shape it was collected on; None otherwise."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_BWD_FILE if kind == "bwd" else PMC_FILE)))
        sh = pmc["shape"]
        same = all(sh[k] == dims[k] for k in ("N", "S", "M", "C", "L", "Lq", "P")) and \
            sh["value_dtype"] == ("bf16" if dims["e"] == 2 else "f32")
        if not same:
            return None
        if kind == "bwd":       # the whole chain
            # every kernel of one backward call (round 4: counting pre-pass, scan, grad_loc / grad_attn + records, the two
            # grad_value walks, row store; the two zero fills are one `zero16` entry averaged over both launches)
            names = [k for k in pmc["kernels"] if k.startswith("bwd_") or k.startswith("cell_") or k.startswith("scan_") or k == "coarse_rows_store"]
            total = sum(pmc["kernels"][k].get("hbm_bytes_per_launch", 0.0) for k in names)
            total += 2.0 * pmc["kernels"].get("zero16", {}).get("hbm_bytes_per_launch", 0.0)
            return round(total / 1e6, 1)
        return round(pmc["kernels"]["fwd_pcm"]["hbm_bytes_per_launch"] / 1e6, 1)
    except Exception:
        return None


def gather_other_states(dev, batch, iters=10):
    """The forward gather alone at the step's operator shape on locations that are NOT the initial state the timed step
    runs on (random-init weights put every sampling offset on whole voxels; a trained model does not sit there): the refine
    block's pattern with +-0.3 voxel of jitter, and ops/test.py's uniform locations (no locality at all).  hipEvent pairs
    of the library on the launch stream, as for the step's own launches; -> {name: {avg_launch_ms, frac}}."""
    import torch
    from transoar_amd import MSDA, _native
    shapes_l = [(40, 40, 64), (20, 20, 32), (10, 10, 16), (5, 5, 8)]
    M, C, L, P = 6, 64, 4, 4
    shapes = torch.as_tensor(shapes_l, dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    g = torch.Generator(device=dev).manual_seed(7)
    ref = []
    for D, H, W in shapes_l:                      # voxel centres, (x, y, z)
        z, y, x = torch.meshgrid(torch.arange(D, device=dev), torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
        ref.append(torch.stack(((x.reshape(-1) + 0.5) / W, (y.reshape(-1) + 0.5) / H, (z.reshape(-1) + 0.5) / D), -1))
    ref = torch.cat(ref, 0).float()
    dirs = torch.tensor([(-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 1), (0, 1, 0), (1, 0, 0)], dtype=torch.float32, device=dev)
    off = (dirs[:, None, None, :] * torch.arange(1, P + 1, dtype=torch.float32, device=dev)[None, None, :, None]).expand(M, L, P, 3)
    off = off + 0.6 * (torch.rand(batch, S, M, L, P, 3, device=dev, generator=g) - 0.5)
    loc_model = (ref[None, :, None, None, None, :] + off / shapes.flip(-1).float()[None, None, None, :, None, :]).contiguous()
    value = torch.randn(batch, S, M, C, device=dev, generator=g).to(torch.bfloat16)
    attn = torch.softmax(torch.randn(batch, S, M, L * P, device=dev, generator=g), -1).view(batch, S, M, L, P)
    nbytes = msda_algorithmic_bytes("fwd", N=batch, S=S, M=M, C=C, L=L, Lq=S, P=P, e=2, e_loc=4)
    out = {}
    for name, loc in (("jitter_0.3_voxel", loc_model), ("uniform", torch.rand(loc_model.shape, device=dev, generator=g))):
        for _ in range(2):
            MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, attn, 64)
        torch.cuda.synchronize()
        _native.profile_enable(True)
        _native.profile_read()
        for _ in range(iters):
            MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, attn, 64)
        torch.cuda.synchronize()
        _native.profile_enable(False)
        ms, n = _native.profile_read()["fwd"]
        out[name] = {"avg_launch_ms": round(ms / n, 4), "frac": round(nbytes / (ms / n) / 1e6 / HBM_PEAK_GBPS, 4)}
    return out


def cpu_baseline_leg():
    """Runs in a subprocess on the host CPU.  Bounded sample of the workload the metric times: ONE whole training step
    (forward + criterion + backward + AdamW) of the flagship model on its use_cuda=False path -- TransoarNet with the
    oracle's torch restatement of ms_deform_attn_core_pytorch (grid_sample) injected -- fp32, batch 1 at the flagship
    geometry, refine on, torch.set_num_threads(min(host cores, 64)) (the best count of round 3's sweep): one untimed
    forward (lazy initialisation, allocator warm-up), then one timed step, ~30 s.  `value` = 1 / that time, in the
    metric's unit.  Beside it, in SECONDS PER CALL (round-4 VERDICT weak #13: the operator alone is not a volumes/s
    figure): the hot operator, MSDeformAttn forward and forward+backward at N=1 S=Lq=117000 M=6 C=64 L=4 P=4, one warm
    + one timed call.  The 3-iteration protocol of SURVEY 8d is `--cpu-baseline-step` (profiles/r05_cpu_step.json)."""
    import torch
    from oracle.torch_ref import msda3d_core_torch
    from tests._inputs import VISCERAL_LEVELS, model_like_inputs
    from transoar_amd import ms_deform_attn
    from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config
    from transoar_amd.train_step import TrainStep
    from transoar_amd.transoarnet import TransoarNet, build_criterion
    cores = os.cpu_count()
    threads = int(os.environ.get("TRANSOAR_CPU_STEP_THREADS", str(min(cores, 64))))
    torch.set_num_threads(threads)
    ms_deform_attn.register_debug_core(msda3d_core_torch)
    cfg = visceral_config(refine=True, use_cuda=False)
    cfg["bbox_properties"] = synthetic_bbox_properties(cfg["num_classes"], seed=0)
    torch.manual_seed(0)
    model = TransoarNet(cfg)
    step = TrainStep(model, build_criterion(cfg), cfg, amp_dtype=torch.float32)
    x = torch.rand(1, 1, *cfg["volume_shape"], generator=torch.Generator().manual_seed(1234))
    targets = synthetic_targets(1, cfg["num_classes"], seed=1)
    model.train()
    t0 = time.perf_counter()
    with torch.no_grad():
        step.loss(x, targets)
    t_warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    step(x, targets)
    t_step = time.perf_counter() - t0
    del step, model

    value, shapes, lsi, loc, attn = model_like_inputs(0, 1, VISCERAL_LEVELS)
    value.requires_grad_(); loc.requires_grad_(); attn.requires_grad_()

    def one():
        for t in (value, loc, attn):
            t.grad = None
        t0 = time.perf_counter()
        out = msda3d_core_torch(value, shapes, loc, attn)
        t_fwd = time.perf_counter() - t0
        out.backward(torch.ones_like(out))
        return t_fwd, time.perf_counter() - t0

    one()
    op_fwd, op_fwd_bwd = one()
    print(json.dumps({"value": round(1.0 / t_step, 5), "unit": "volumes/s", "cores": threads, "kind": "port", "host_cores": cores,
                      "step_s": round(t_step, 2), "untimed_forward_s": round(t_warm, 2),
                      "operator_s_per_call": {"forward": round(op_fwd, 2), "forward_backward": round(op_fwd_bwd, 2)},
                      "sample": "ONE whole training step (fwd + criterion + bwd + AdamW) of the flagship model, use_cuda=False with "
                                "the oracle torch core, fp32, batch 1 (one 160x160x256 volume), refine on, %d threads of %d host "
                                "cores; one untimed forward first, then the timed step: %.1f s.  operator_s_per_call: the "
                                "MSDeformAttn core alone at N=1 S=Lq=117000 M=6 C=64 L=4 P=4 fp32 (2 calls per volume), one warm + "
                                "one timed call" % (threads, cores, t_step)}))


def cpu_step_leg():
    """The model's use_cuda=False path as a whole (SURVEY 8d protocol): TransoarNet with the oracle's torch
    restatement of ms_deform_attn_core_pytorch injected, fp32, batch 1 at the flagship geometry, refine on,
    every host core; 1 warm + 3 timed iterations, median; forward-only and the full training step.  Minutes per
    iteration: run separately (python bench.py --cpu-baseline-only --cpu-baseline-step), the result is kept in
    profiles/r05_cpu_step.json and quoted by the default run as `cpu_step`."""
    import torch
    from oracle.torch_ref import msda3d_core_torch
    from transoar_amd import ms_deform_attn
    from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config
    from transoar_amd.train_step import TrainStep
    from transoar_amd.transoarnet import TransoarNet, build_criterion
    cores = os.cpu_count()
    threads = int(os.environ.get("TRANSOAR_CPU_STEP_THREADS", str(min(cores, 64))))     # round 2 used all 256: oversubscribed
    torch.set_num_threads(threads)
    ms_deform_attn.register_debug_core(msda3d_core_torch)
    cfg = visceral_config(refine=True, use_cuda=False)
    cfg["bbox_properties"] = synthetic_bbox_properties(cfg["num_classes"], seed=0)
    torch.manual_seed(0)
    model = TransoarNet(cfg)
    step = TrainStep(model, build_criterion(cfg), cfg, amp_dtype=torch.float32)
    x = torch.rand(1, 1, *cfg["volume_shape"], generator=torch.Generator().manual_seed(1234))
    targets = synthetic_targets(1, cfg["num_classes"], seed=1)
    timed = int(os.environ.get("TRANSOAR_CPU_STEP_ITERS", "3"))

    def median_of(fn):
        fn()                                    # 1 warm
        ts = []
        for _ in range(timed):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2], ts

    model.train()
    def fwd_only():
        with torch.no_grad():
            step.loss(x, targets)
    t_fwd, all_fwd = median_of(fwd_only)
    t_step, all_step = median_of(lambda: step(x, targets))
    rec = {"value": round(1.0 / t_step, 5), "unit": "volumes/s", "cores": threads, "host_cores": cores, "kind": "port",
           "forward_only_volumes_per_s": round(1.0 / t_fwd, 5), "step_s": [round(t, 2) for t in all_step],
           "forward_s": [round(t, 2) for t in all_fwd],
           "sample": "whole training step (fwd + criterion + bwd + AdamW) of the flagship model, use_cuda=False with the "
                     "oracle torch core, fp32, batch 1, refine on, torch.set_num_threads(%d) of %d host cores; 1 warm + %d timed, median" % (threads, cores, timed)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "cpu_step_%dthreads.json" % threads), "w"), indent=1)
    print(json.dumps(rec))


def cpu_step_record():
    """The committed 3-iteration whole-step CPU measurement (profiles/r05_cpu_step.json), quoted with its provenance."""
    for name in ("r05_cpu_step.json", "r03_cpu_step.json"):
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        rec["provenance"] = "measured separately on an MI355X box's host by `bench.py --cpu-baseline-only --cpu-baseline-step`, " \
                            "profiles/%s; NOT re-measured in this run (`cpu_baseline` is this run's own one-step sample)" % name
        return rec
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2, help="volumes per GPU (config batch_size)")
    ap.add_argument("--no-refine", action="store_true", help="shipped default: use_decoder_attn=False")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole step (forward + loss + backward [+ all-reduce] + AdamW) as one HIP graph (host enqueue "
                         "3 ms per step instead of 17).  Default on ONE GPU: both modes are paced during the warm-up and the "
                         "faster one is timed (they are within 1.5 %% of each other, which one leads depends on the host: "
                         "DESIGN.md section 12.4); on more than one GPU the eager step, whose bucketed all-reduce overlaps the "
                         "backward (sections 6, 8)")
    ap.add_argument("--swin", action="store_true", help="BASELINE config #4: Swin encoder stages (use_encoder_attn=True)")
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="the eager step, without pacing the graph replay first")
    ap.add_argument("--miopen-benchmark", action="store_true", help="torch.backends.cudnn.benchmark=True")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--cpu-baseline-step", action="store_true",
                    help="with --cpu-baseline-only: time the WHOLE use_cuda=False training step of the model on all host "
                         "cores (1 warm + 3 timed, forward-only and full step; minutes per step) instead of the bounded "
                         "operator sample; prints the record bench.py's cpu_step field quotes (profiles/r05_cpu_step.json)")
    ap.add_argument("--cpu-baseline-timeout", type=float, default=300.0)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        return cpu_step_leg() if args.cpu_baseline_step else cpu_baseline_leg()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if rank != 0:
        os.dup2(2, 1)          # only rank 0 owns stdout (library banners of the other ranks go to stderr)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    torch.backends.cudnn.benchmark = bool(args.miopen_benchmark)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    elif os.environ.get("TRANSOAR_FORCE_DP"):
        # test hook: a ONE-rank RCCL group, so that the whole data-parallel path (communicator, bucketed
        # all-reduce, rank-summed loss normalisers, graph capture next to the watchdog thread) can be
        # exercised on a single-GPU box
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=dev)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from transoar_amd import _native
    from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config
    from transoar_amd.matcher import DenseTargets
    from transoar_amd.train_step import TrainStep
    from transoar_amd.transoarnet import TransoarNet, build_criterion

    cfg = visceral_config(refine=not args.no_refine, use_cuda=True, swin=args.swin)
    cfg["bbox_properties"] = synthetic_bbox_properties(cfg["num_classes"], seed=0)
    torch.manual_seed(0)                       # identical replicas
    model = TransoarNet(cfg).to(dev)
    amp = torch.float32 if args.fp32 else torch.bfloat16
    # One GPU, no flag: capture the step, pace graph replay and eager step for 8 steps each during the warm-up and time the
    # faster one (round 6: with the synchronous copy gone from the optimizer the host needs 16-19 ms to enqueue a 32-ms step,
    # and the eager step measured 0.1-0.45 ms FASTER than the replay on every box of the round; a slower or busier host turns
    # that around, and the replay needs 3 ms of host time per step).  More than one GPU, no flag: the eager step.
    auto = not args.graph and not args.no_graph and world == 1
    if not args.graph and not auto:
        args.no_graph = True
    step = TrainStep(model, build_criterion(cfg), cfg, amp_dtype=amp, graph=not args.no_graph)

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.rand(args.batch, 1, *cfg["volume_shape"], device=dev, generator=g)
    targets = DenseTargets.from_list(synthetic_targets(args.batch, cfg["num_classes"], seed=1 + rank, device=dev),
                                     cfg["num_classes"], dev)

    def trace(msg):
        if os.environ.get("TRANSOAR_BENCH_TRACE"):
            torch.cuda.synchronize()
            print("[bench] %s" % msg, file=sys.stderr, flush=True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step_mode = "eager"
    if args.no_graph:
        step(x, targets)                       # one eager step first (lazy init, MIOpen find-db lookups)
    else:                                      # ... on the stream the capture will use (TrainStep.capture)
        side = step.capture_stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step(x, targets)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
    trace("first eager step done")
    if not args.no_graph:
        try:
            step.capture(x, targets)
            step(x, targets)                   # first replay: TrainStep checks its loss against the eager step's
            step_mode = ("hip-graph(fwd+loss+bwd+all-reduce+AdamW)" if getattr(step, "capture_exchange", False) and step.capture_optimizer else
                         "hip-graph(fwd+loss+bwd+AdamW)" if step.capture_optimizer else
                         "hip-graph(fwd+loss+bwd) + eager all-reduce + fused AdamW")
        except Exception as e:                 # keep going eagerly, say so in the output
            import traceback
            traceback.print_exc()
            step_mode = "eager (graph capture failed: %s)" % (str(e).splitlines()[0][:120],)
            step.drop_graph()
    trace("capture done: %s" % step_mode)
    calibration = None
    if auto and step._graph is not None:
        def pace(n=8):
            # like the timed region: hipEvents on the compute stream around n steps, after three steps that let the host get
            # ahead of the GPU (from a drained queue the eager step starves in its launch-bound stretches for about one step:
            # a first version that took wall clock from a synchronize charged the eager mode 1 ms per step for it)
            for _ in range(3):
                step(x, targets)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                step(x, targets)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / n
        t_graph = pace()
        graph, step._graph = step._graph, None
        step.reducer.overlap = True
        # the eager steps of a TrainStep that holds a graph run on its capture stream; with the whole loop on that stream there
        # is no hand-over per step (the two stream waits cost 0.9 ms per step: the queue drains at every step boundary)
        side = step.capture_stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            t_eager = pace()
        torch.cuda.current_stream().wait_stream(side)
        calibration = {"graph_ms": round(t_graph, 3), "eager_ms": round(t_eager, 3), "steps_each": 8,
                       "note": "hipEvent time per step over 8 steps of each mode during the warm-up (after 3 ramp-up steps); the faster mode is the one timed"}
        if t_eager <= t_graph:
            step.drop_graph()
            step_mode = "eager"
            import contextlib
            on_side = contextlib.ExitStack()               # everything from here on runs on the capture stream
            on_side.enter_context(torch.cuda.stream(side))
        else:
            step._graph = graph
            step.reducer.overlap = bool(getattr(step, "capture_exchange", False))
        trace("paced: graph %.3f ms, eager %.3f ms -> %s" % (t_graph, t_eager, step_mode))
    for i in range(args.warmup):
        step(x, targets)
        trace("warmup %d" % i)
    barrier()
    t0 = time.perf_counter()
    host_s = 0.0
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # hipEvents on the compute stream
    marks[0].record()
    for i in range(args.steps):
        h0 = time.perf_counter()
        total, _ = step(x, targets)
        host_s += time.perf_counter() - h0          # time the host needs to ENQUEUE a step (no sync)
        marks[i + 1].record()
        if os.environ.get("TRANSOAR_BENCH_TRACE"):
            if os.environ.get("TRANSOAR_BENCH_GC"):
                import gc
                gc.collect()
            trace("timed step loss %.4f" % float(total))
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    trace("timed region done")
    # what the host needs to enqueue ONE step into an EMPTY queue (host_enqueue_ms_per_step above is taken with the GPU running
    # behind: once the queue is full it measures back-pressure, i.e. the step time again).  The mode that was timed here; the eager
    # step (the data-parallel default, one process per GPU: what 8 ranks on one host contend with) below, among the eager
    # profile steps.
    def drained(n=4):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            step(x, targets)
            ts.append(time.perf_counter() - h0)
        torch.cuda.synchronize()
        return round(sorted(ts[1:])[(n - 1) // 2] * 1e3, 2)
    host_drained = {"timed_mode": drained()}
    # per-kernel durations of the MSDeformAttn kernels: hipEvent pairs recorded by the library on the launch stream, over
    # three eager steps of the same model state run right after the timed region (the timed steps themselves carry no
    # profiling events; a replayed graph re-records nothing anyway; kernel durations do not depend on how the launch was issued)
    graph, step._graph = step._graph, None
    if graph is not None:
        step.reducer.overlap = True
    prof_steps = 3
    _native.profile_enable(True)
    _native.profile_read()
    for i in range(prof_steps):
        step(x, targets)
        trace("eager profile step %d" % i)
    torch.cuda.synchronize()
    _native.profile_enable(False)
    prof = _native.profile_read()
    host_drained["eager"] = drained() if graph is not None else host_drained["timed_mode"]
    step._graph = graph
    other_states = None
    if rank == 0 and not args.no_refine and not args.fp32 and not args.swin and not os.environ.get("TRANSOAR_BENCH_SKIP_OTHER"):
        try:
            other_states = gather_other_states(dev, args.batch)
        except Exception as e:          # report, never fake
            other_states = {"error": repr(e)[:200]}
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_value = float(total)
    assert loss_value == loss_value, "loss is NaN"

    if rank == 0:
        global_batch = args.batch * world
        # MSDeformAttn problem of the refine block at this geometry
        shapes = [(40, 40, 64), (20, 20, 32), (10, 10, 16), (5, 5, 8)]
        S = sum(d * h * w for d, h, w in shapes)
        dims = dict(N=args.batch, S=S, M=6, C=64, L=4, Lq=S, P=4, e=4 if args.fp32 else 2, e_loc=4)
        kernels = {}
        for kind, (ms, n) in prof.items():
            if n:
                kernels[kind] = {"launches_per_step": n / prof_steps, "avg_ms": round(ms / n, 4)}
        timing = "hipEvent pairs on the launch stream, 3 eager steps right after the timed region"
        roofline = msda_bwd = None
        if "fwd" in kernels:                    # the MSDeformAttn forward gather: the kernel north_star names
            b = msda_algorithmic_bytes("fwd", **dims)
            gbps = b / kernels["fwd"]["avg_ms"] / 1e6
            roofline = {"kernel": "msda3d_fwd_pcm", "bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": pmc_traffic("fwd", dims),
                        "traffic_unit": "MB per launch (PMC, profiles/%s: collected on the launches of this training step, eager mode)" % PMC_FILE,
                        "avg_launch_ms": kernels["fwd"]["avg_ms"], "algorithmic_MB": round(b / 1e6, 1), "timing": timing,
                        "state": "initial state (random-init weights: every sampling offset a whole number of voxels) -- the best case",
                        "other_states": other_states}
        chain = [k for k in BWD_KINDS if k in kernels]
        if chain:                               # one backward call = the chain of these kernels, against B_bwd
            calls = kernels["bwd_query"]["launches_per_step"] if "bwd_query" in kernels else kernels[chain[0]]["launches_per_step"]
            ms_call = sum(kernels[k]["avg_ms"] * kernels[k]["launches_per_step"] for k in chain) / calls
            b = msda_algorithmic_bytes("bwd", **dims)
            msda_bwd = {"kernels": chain, "ms_per_call": round(ms_call, 4), "algorithmic_MB": round(b / 1e6, 1),
                        "achieved_GBps": round(b / ms_call / 1e6, 1), "frac": round(b / ms_call / 1e6 / HBM_PEAK_GBPS, 4),
                        "traffic": pmc_traffic("bwd", dims), "traffic_unit": "MB per call (PMC, profiles/%s: the training step's own launches)" % PMC_BWD_FILE}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"],
                                   capture_output=True, text=True, timeout=args.cpu_baseline_timeout,
                                   env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
                cpu = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:    # report, never fake
                cpu = {"value": None, "unit": "volumes/s", "cores": os.cpu_count(), "kind": "port",
                       "sample": "cpu baseline leg failed: %r" % (e,)}
        line = json.dumps({
            "metric": "training-step CT volumes/s, 160x160x256 Focused-Decoder",
            "value": round(global_batch * args.steps / elapsed, 4), "unit": "volumes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "step_ms": {"median": round(step_ms[len(step_ms) // 2], 3), "p10": round(step_ms[int(0.1 * (len(step_ms) - 1))], 3),
                        "p90": round(step_ms[int(0.9 * (len(step_ms) - 1))], 3),
                        "timing": "hipEvent per step on the compute stream (this rank)"},
            "vs_baseline": None, "dtype": "f32" if args.fp32 else "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: Focused Decoder, 160x160x256 (VISCERAL geometry), "
                                   "batch 2 per GPU, bf16 autocast, refine %s%s" % ("off" if args.no_refine else "on (use_decoder_attn, use_cuda)",
                                                                       ", Swin encoder (configs[3])" if args.swin else ""),
                       "global_batch": global_batch, "per_gpu_batch": args.batch, "volume": list(cfg["volume_shape"]),
                       "parallelism": "dp%d" % world, "weights": "random init", "optimizer": "AdamW fused",
                       "step_mode": step_mode, "step_mode_pacing": calibration,
                       "params": sum(p.numel() for p in model.parameters())},
            "loss": round(loss_value, 5), "host_enqueue_ms_per_step": round(host_s / args.steps * 1e3, 2),
            "host_enqueue_drained_ms": dict(host_drained, note="host time to enqueue one step into an EMPTY queue (median of 3 after a "
                                            "synchronize); host_enqueue_ms_per_step is taken with the GPU running behind and tracks the step time"),
            "roofline": roofline, "msda_backward": msda_bwd, "msda_kernels": kernels, "cpu_baseline": cpu,
            "parity": {"normalisation": "every tolerance of tests/ is TENSOR-MAX normalised: max|a - b| / max|b| "
                                        "(north_star's 'max rel-err' read that way); bounds: fp32 1e-4, fp64 1e-10, bf16 / f16 storage "
                                        "2^-7 / 2^-10",
                       "elementwise": "additionally, on the entries above 1 % of the tensor maximum, |a - b| / |b| <= 1e-4 (fp32), "
                                      "2^-6 / 2^-9 (bf16 / f16 storage; 2^-7 for the full-size bf16 out against the C oracle: the "
                                      "output rounding) for out / grad_value / grad_loc / grad_attn (tests/test_msda_gpu.py: "
                                      "ELEM_TOL, FULL_ELEM_OUT, elem_relerr)",
                       "oracle": "oracle/ (C + torch restatements), pinned to tests/golden/g1-g9 generated by importing the reference"},
            "cpu_step": cpu_step_record(),
        })
    else:
        line = None
    if dist.is_initialized():
        dist.destroy_process_group()
    # RCCL printf()s a banner ("Librccl path : ...") into C stdio, which a pipe holds back until the
    # process exits, i.e. AFTER Python's own output: drain it first so the JSON line is the last line
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if line is not None:
        print(line, flush=True)


if __name__ == "__main__":
    main()
