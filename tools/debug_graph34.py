"""What corrupts the captured refine-on training step at its 34th replay?  (dev tool; the answer is the
runtime's graph packet capture: DESIGN.md section 8)
   python tools/debug_graph34.py MODE [packet_capture]
MODE: none | skip_bwd | skip_bwd_loc | skip_bwd_value | no_dropout | no_refine_dropout | gradnorm | rocblas |
      side_first | no_first | accum | accum_eager | poison | poison_empty
"packet_capture" as second argument runs with DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 (the runtime's default),
which reproduces the corruption; without it the script switches it off like the package does."""
import os, sys
if "packet_capture" in sys.argv[1:]:          # the runtime's default: reproduces the corruption
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "1"
    os.environ["TRANSOAR_TRUST_PACKET_CAPTURE"] = "1"
else:
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, "miopen_db"))
import transoar_amd.ms_deform_attn as mda
from transoar_amd import msda as MSDA
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
orig = MSDA.ms_deform_attn_backward
def patched(value, shapes, starts, loc, attn, go, step):
    if mode == "skip_bwd":
        return [torch.zeros_like(value), torch.zeros_like(loc), torch.zeros_like(attn)]
    gv, gl, ga = orig(value, shapes, starts, loc, attn, go, step)
    if mode == "skip_bwd_loc":
        return [gv, torch.zeros_like(gl), torch.zeros_like(ga)]
    if mode == "skip_bwd_value":
        return [torch.zeros_like(gv), gl, ga]
    return [gv, gl, ga]
mda.MSDA.ms_deform_attn_backward = patched
from transoar_amd.config import synthetic_bbox_properties, synthetic_targets, visceral_config
from transoar_amd.matcher import DenseTargets
from transoar_amd.train_step import TrainStep
from transoar_amd.transoarnet import TransoarNet, build_criterion
if mode == "rocblas":
    torch.backends.cuda.preferred_blas_library("cublas")      # rocBLAS instead of hipBLASLt
cfg = visceral_config(refine=True, use_cuda=True)
cfg["bbox_properties"] = synthetic_bbox_properties(cfg["num_classes"], seed=0)
torch.manual_seed(0)
model = TransoarNet(cfg).cuda()
if mode == "no_dropout":
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
if mode == "no_refine_dropout":
    for m in model._backbone._decoder._refine.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
step = TrainStep(model, build_criterion(cfg), cfg, amp_dtype=torch.bfloat16, graph=True)
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.rand(2, 1, *cfg["volume_shape"], device="cuda", generator=g)
targets = DenseTargets.from_list(synthetic_targets(2, cfg["num_classes"], seed=1, device="cuda"), cfg["num_classes"], "cuda")
if mode in ("poison", "poison_empty"):
    step(x, targets)
elif mode == "side_first":
    side = step.capture_stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step(x, targets)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
elif mode == "no_first":
    pass
else:
    step(x, targets)
if mode not in ("poison", "poison_empty"):
    step.capture(x, targets)
if mode == "poison_empty":
    # every torch.empty/empty_like/new_empty made by the Python side of the step comes back filled with
    # NaN (floats) or 0x7f bytes (integers): a kernel that leaves part of its output unwritten, or reads
    # scratch before writing it, turns into non-finite gradients
    import transoar_amd
    def _poison(t):
        if t.is_cuda and t.numel():
            if t.is_floating_point():
                t.fill_(float("nan"))
            else:
                t.view(torch.uint8).fill_(0x7f) if t.is_contiguous() else t.fill_(1 << 20)
        return t
    _e, _el, _ne, _es = torch.empty, torch.empty_like, torch.Tensor.new_empty, torch.empty_strided
    torch.empty = lambda *a, **k: _poison(_e(*a, **k))
    torch.empty_like = lambda *a, **k: _poison(_el(*a, **k))
    torch.empty_strided = lambda *a, **k: _poison(_es(*a, **k))
    torch.Tensor.new_empty = lambda self, *a, **k: _poison(_ne(self, *a, **k))
    params = dict(model.named_parameters())
    step._graph = None
    for k in range(2):
        step._eager_fwd_bwd(x, targets); torch.cuda.synchronize()
        bad = [n for n, p in params.items() if p.grad is not None and not torch.isfinite(p.grad).all()]
        print("poisoned-empty eager step", k, "non-finite grads:", len(bad), bad[:40], flush=True)
    sys.exit(0)
if mode == "poison":
    # eager step on NaN-poisoned free memory: any read of an unwritten torch.empty buffer shows up as a
    # non-finite gradient
    params = dict(model.named_parameters())
    step._graph = None
    for k in range(2):
        junk = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(24)]     # 24 GiB of NaN
        del junk
        step._eager_fwd_bwd(x, targets); torch.cuda.synchronize()
        bad = [n for n, p in params.items() if p.grad is not None and not torch.isfinite(p.grad).all()]
        print("poisoned eager step", k, "non-finite grads:", len(bad), bad[:12], flush=True)
    sys.exit(0)
if mode == "accum_eager":
    params = dict(model.named_parameters())
    probe = ["_backbone._decoder._refine.refine_def_attn.layers.0.linear1.weight", "_backbone._encoder._stages.0._block.0.weight", "_neck.decoder.layers.0.linear1.weight", "_cls_head.weight"]
    for k in range(3):
        step._graph = None
        step._eager_fwd_bwd(step._static_x, step._static_t); torch.cuda.synchronize()
        print("eager fwd+bwd without optimizer", k, ["%.4e" % float(params[n].grad.float().norm()) for n in probe], flush=True)
    sys.exit(0)
if mode == "accum":
    params = dict(model.named_parameters())
    probe = ["_backbone._decoder._refine.refine_def_attn.layers.0.linear1.weight", "_backbone._encoder._stages.0._block.0.weight", "_neck.decoder.layers.0.linear1.weight", "_cls_head.weight"]
    for k in range(3):
        step._graph.replay(); torch.cuda.synchronize()
        print("replay without optimizer", k, ["%.4e" % float(params[n].grad.float().norm()) for n in probe], flush=True)
        if k == 0:
            first = {n: p.grad.detach().float().clone() for n, p in params.items() if p.grad is not None}
        if k == 1:
            # which parameters moved between the first and the second replay (same inputs, same weights;
            # only the dropout masks differ), in registration order = roughly reverse backward order
            for n, g0 in first.items():
                g1 = params[n].grad.detach().float()
                rel = float((g1 - g0).norm() / (g0.norm() + 1e-30))
                print("  %-78s |g0| %.3e  rel diff %.3e" % (n, float(g0.norm()), rel), flush=True)
    sys.exit(0)
names = ["_backbone._decoder._refine.level_embed", "_backbone._decoder._refine.refine_def_attn.layers.0.linear1.weight",
         "_backbone._decoder._refine.refine_def_attn.layers.1.norm2.weight", "_backbone._encoder._stages.0._block.0.weight",
         "_neck.decoder.layers.0.linear1.weight", "_backbone._decoder._out.0.weight"]
params = dict(model.named_parameters())
for i in range(45):
    t, _ = step(x, targets)
    torch.cuda.synchronize()
    v = float(t)
    if mode == "gradnorm" and (i < 6 or i > 28):
        print(i + 1, "loss %.4f" % v, ["%.3e" % float(params[n].grad.float().norm()) for n in names], ["%.3e" % float(params[n].float().abs().max()) for n in names[:3]], flush=True)
    if v != v:
        print(mode, "NaN at replay", i + 1); break
else:
    print(mode, "45 replays fine, last loss %.4f" % v)
