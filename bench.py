#!/usr/bin/env python
"""Headline benchmark: training-step CT volumes/s of the Focused-Decoder model
on synthetic 160x160x256 volumes (BASELINE.json metric), 1..8 MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
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
  cpu_baseline  one training step of the same model on the host CPU cores with
                the oracle's torch core (kind "port"), batch 1, fp32
"""
import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # see transoar_amd/__init__.py; before torch loads HIP

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The torch-side convolutions (everything that is not yet a hand-written kernel) go
# through MIOpen, whose untuned fallback for these 3-D bf16 shapes is a naive
# reference kernel (1.5 s per step for one layer).  miopen_db/ holds the user
# find-db tuned once on an MI355X (python bench.py --miopen-benchmark with
# MIOPEN_USER_DB_PATH pointing there); using it is plumbing, not product.
_DB = os.path.join(ROOT, "miopen_db")
if os.path.isdir(_DB) and "MIOPEN_USER_DB_PATH" not in os.environ:
    os.environ["MIOPEN_USER_DB_PATH"] = _DB

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def msda_algorithmic_bytes(kind, N, S, M, C, L, Lq, P, e, e_loc, fine=None):
    """Bytes one launch must move if every tensor is touched once."""
    value, out = e * N * S * M * C, e * N * Lq * M * C
    loc_attn = e_loc * N * Lq * M * L * P * 4
    points = N * Lq * M * L * P
    recs = points * (32 + 4)                                  # sorted 8-weight record + grad_out row index
    fine = L if fine is None else fine                        # levels accumulated per brick in LDS (the rest: chunked walk)
    return {
        "fwd": value + out + loc_attn,                       # SURVEY 8d B_fwd
        "bwd_query": value + out + 2 * loc_attn + 4 * points,    # value, grad_out in; grad_loc/attn + ranks out
        "pull": out + value + loc_attn,                      # fallback: grad_out in, grad_value out, point geometry
        # fine levels: every sorted point + its grad_out row once, grad_value rows out.  The figure is
        # for the whole pyramid (the coarse levels' share of points is in value_cells), split by level:
        "value_tile": (recs + out) * fine // L + value,
        "value_cells": (recs + out) * (L - fine) // L,
        "cell_count": loc_attn, "cell_fill": loc_attn + recs, "scan": 0,
    }.get(kind, 0)


PMC_FILE = "r01_msda_pmc_v3.json"


def pmc_traffic(kind, dims):
    """HBM bytes per launch from the committed PMC collection (profiles/<PMC_FILE>:
    FETCH_SIZE/WRITE_SIZE, separate rocprofv3 --pmc passes, gfx950 correction applied), valid only
    for the shape it was collected on; None otherwise."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
        sh = pmc["shape"]
        same = all(sh[k] == dims[k] for k in ("N", "S", "M", "C", "L", "Lq", "P")) and \
            sh["value_dtype"] == ("bf16" if dims["e"] == 2 else "f32")
        name = {"fwd": "fwd_brick", "bwd_query": "bwd_query_brick", "value_tile": "bwd_value_tile",
                "value_cells": "bwd_value_cells", "cell_fill": "cell_fill_w8"}[kind]
        return round(pmc["kernels"][name]["hbm_bytes_per_launch"] / 1e6, 1) if same else None
    except Exception:
        return None


def cpu_baseline_leg():
    """Runs in a subprocess on the host CPU.  Bounded sample: the hot operator of the step -- the
    refine block's MSDeformAttn forward+backward at the flagship geometry (N=1, S=Lq=117000, M=6,
    C=64, L=4, P=4, fp32) through the oracle's torch restatement of the reference's
    use_cuda=False core (grid_sample).  A volume needs it twice (2 refine layers), so the figure is
    1 / (2 * t) volumes/s of the operator path alone; the rest of the step is NOT included (a whole
    CPU step of this model took 269 s on the 256-core host of the round-1 GPU box)."""
    import torch
    from oracle.torch_ref import msda3d_core_torch
    from tests._inputs import VISCERAL_LEVELS, model_like_inputs
    cores = os.cpu_count()
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    value, shapes, lsi, loc, attn = model_like_inputs(0, 1, VISCERAL_LEVELS)
    value.requires_grad_(); loc.requires_grad_(); attn.requires_grad_()
    t0 = time.perf_counter()
    out = msda3d_core_torch(value, shapes, loc, attn)
    t_fwd = time.perf_counter() - t0
    out.backward(torch.ones_like(out))
    dt = time.perf_counter() - t0
    print(json.dumps({"value": round(1.0 / (2 * dt), 5), "unit": "volumes/s", "cores": threads, "kind": "port",
                      "sample": "MSDeformAttn fwd+bwd only (the hot operator; 2 calls per volume), N=1 flagship "
                                "shape S=Lq=117000 M=6 C=64 L=4 P=4, fp32, oracle torch core (grid_sample), "
                                "%d threads of %d host cores: fwd %.2f s, fwd+bwd %.2f s per call; the rest of the "
                                "training step is not in this figure" % (threads, cores, t_fwd, dt)}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2, help="volumes per GPU (config batch_size)")
    ap.add_argument("--no-refine", action="store_true", help="shipped default: use_decoder_attn=False")
    ap.add_argument("--graph", action="store_true",
                    help="replay forward+loss+backward as one HIP graph: the default on one GPU (about 2 ms faster per "
                         "step at K=10, host enqueue 9 ms instead of 56 ms).  With more than one rank the default is the "
                         "eager step, whose bucketed all-reduce overlaps the backward (DESIGN.md sections 6 and 8)")
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager step also on one GPU")
    ap.add_argument("--miopen-benchmark", action="store_true", help="torch.backends.cudnn.benchmark=True")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--cpu-baseline-timeout", type=float, default=240.0)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        return cpu_baseline_leg()

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

    cfg = visceral_config(refine=not args.no_refine, use_cuda=True)
    cfg["bbox_properties"] = synthetic_bbox_properties(cfg["num_classes"], seed=0)
    torch.manual_seed(0)                       # identical replicas
    model = TransoarNet(cfg).to(dev)
    amp = torch.float32 if args.fp32 else torch.bfloat16
    if not args.graph and (world > 1 or dist.is_initialized()):
        args.no_graph = True                   # data-parallel default: eager step with the overlapped exchange
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
            step_mode = "hip-graph(fwd+loss+bwd) + eager all-reduce + fused AdamW"
        except Exception as e:                 # keep going eagerly, say so in the output
            import traceback
            traceback.print_exc()
            step_mode = "eager (graph capture failed: %s)" % (str(e).splitlines()[0][:120],)
            step.drop_graph()
    trace("capture done: %s" % step_mode)
    for i in range(args.warmup):
        step(x, targets)
        trace("warmup %d" % i)
    barrier()
    if step._graph is None:                    # eager: time the kernels over the timed steps themselves
        _native.profile_enable(True)
        _native.profile_read()
    t0 = time.perf_counter()
    host_s = 0.0
    for _ in range(args.steps):
        h0 = time.perf_counter()
        total, _ = step(x, targets)
        host_s += time.perf_counter() - h0          # time the host needs to ENQUEUE a step (no sync)
        if os.environ.get("TRANSOAR_BENCH_TRACE"):
            if os.environ.get("TRANSOAR_BENCH_GC"):
                import gc
                gc.collect()
            trace("timed step loss %.4f" % float(total))
    barrier()
    elapsed = time.perf_counter() - t0
    trace("timed region done")
    # per-kernel durations of the MSDeformAttn kernels: hipEvent pairs recorded by the library on the
    # launch stream.  A replayed graph re-records nothing, so in graph mode they come from eager steps of
    # the same model state run right after the timed replays (kernel durations do not depend on how the
    # launch was issued); in eager mode they are taken over the timed steps themselves.
    prof_steps = args.steps
    if step._graph is not None:
        graph, step._graph = step._graph, None
        step.reducer.overlap = True
        prof_steps = 3
        _native.profile_enable(True)
        _native.profile_read()
        for i in range(prof_steps):
            step(x, targets)
            trace("eager profile step %d" % i)
        torch.cuda.synchronize()
        step._graph = graph
    _native.profile_enable(False)
    prof = _native.profile_read()
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
        fine = sum(1 for d, h, w in shapes if S * 4 < 32 * d * h * w)      # the dispatch rule of msda3d.hip (kCoarsePointsPerVoxel)
        dims = dict(N=args.batch, S=S, M=6, C=64, L=4, Lq=S, P=4, e=4 if args.fp32 else 2, e_loc=4, fine=fine)
        kernels = {}
        for kind, (ms, n) in prof.items():
            if n == 0:
                continue
            avg = ms / n
            b = msda_algorithmic_bytes(kind, **dims)
            kernels[kind] = {"launches_per_step": n / prof_steps, "avg_ms": round(avg, 4),
                             "algorithmic_MB": round(b / 1e6, 1),
                             "achieved_GBps": round(b / avg / 1e6, 1) if b else None}
        roofline = None
        if kernels:
            dom = max((k for k in kernels if kernels[k]["achieved_GBps"]),
                      key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches_per_step"])
            kd = kernels[dom]
            roofline = {"kernel": "msda3d_" + dom, "bound": "hbm", "achieved": kd["achieved_GBps"],
                        "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(kd["achieved_GBps"] / HBM_PEAK_GBPS, 4),
                        "traffic": pmc_traffic(dom, dims), "traffic_unit": "MB per launch (PMC, profiles/%s)" % PMC_FILE,
                        "avg_launch_ms": kd["avg_ms"], "algorithmic_MB": kd["algorithmic_MB"],
                        "timing": "hipEvent pairs on the launch stream, " + (
                            "timed steps" if step_mode == "eager" or step_mode.startswith("eager") else
                            "3 eager steps right after the timed graph replays")}
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
            "vs_baseline": None, "dtype": "f32" if args.fp32 else "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: Focused Decoder, 160x160x256 (VISCERAL geometry), "
                                   "batch 2 per GPU, bf16 autocast, refine %s" % ("off" if args.no_refine else
                                                                                     "on (use_decoder_attn, use_cuda)"),
                       "global_batch": global_batch, "per_gpu_batch": args.batch, "volume": list(cfg["volume_shape"]),
                       "parallelism": "dp%d" % world, "weights": "random init", "optimizer": "AdamW fused",
                       "step_mode": step_mode,
                       "params": sum(p.numel() for p in model.parameters())},
            "loss": round(loss_value, 5), "host_enqueue_ms_per_step": round(host_s / args.steps * 1e3, 2),
            "roofline": roofline, "msda_kernels": kernels, "cpu_baseline": cpu,
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
