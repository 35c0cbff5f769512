import sys, os, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
from tests._inputs import analytic_volume, fill_deterministic, small_model_config
from transoar_amd.config import synthetic_targets
from transoar_amd.conv3d import Conv3dK3
from transoar_amd.transoarnet import TransoarNet, build_criterion
for refine in (False, True):
    cfg = small_model_config(refine, use_cuda=True)
    torch.manual_seed(0); net = TransoarNet(cfg)
    with torch.no_grad():
        for p_ in net.parameters():          # the heads start at zero (no gradient reaches the body): un-zero them
            if p_.dim() > 1 and float(p_.abs().max()) == 0:
                torch.nn.init.xavier_uniform_(p_)
    net = net.cuda().train()
    x = torch.rand(1, 1, 160, 160, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
    pass
    targets = synthetic_targets(1, 20, seed=1, device="cuda")
    crit = build_criterion(cfg); coefs = cfg["loss_coefs"]
    grads, outs, lv = {}, {}, {}
    old = Conv3dK3.min_voxels
    recorded, assign = [], crit.matcher.assign
    for mode in ("fp32", "bf16"):
        Conv3dK3.min_voxels = 0 if mode == "bf16" else old
        if mode == "fp32":
            crit.matcher.assign = lambda *a, **k: recorded.append(assign(*a, **k)) or recorded[-1]
        else:
            replay = iter(recorded)
            crit.matcher.assign = lambda *a, **k: next(replay)
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(mode == "bf16")):
            out = net(x); losses = crit(out, targets, None, net._anchors)
            total = sum(v * coefs[k.split("_")[0]] for k, v in losses.items())
        total.backward()
        outs[mode] = {k: out[k].detach().float() for k in ("pred_logits", "pred_boxes")}
        lv[mode] = {k: float(v) for k, v in losses.items()}
        grads[mode] = {n: (None if p.grad is None else p.grad.detach().double().flatten().cpu()) for n, p in net.named_parameters()}
    Conv3dK3.min_voxels = old
    print("refine", refine, "boxes maxabs", float((outs["bf16"]["pred_boxes"]-outs["fp32"]["pred_boxes"]).abs().max()),
          "logits maxabs", float((outs["bf16"]["pred_logits"]-outs["fp32"]["pred_logits"]).abs().max()), "lmax", float(outs["fp32"]["pred_logits"].abs().max()),
          "logits rms", float((outs["bf16"]["pred_logits"]-outs["fp32"]["pred_logits"]).pow(2).mean().sqrt()))
    print(" losses", {k: (round(lv["bf16"][k],5), round(v,5)) for k, v in lv["fp32"].items()})
    rel = sorted((float((grads["bf16"][n]-g).norm()/g.norm()), n) for n, g in grads["fp32"].items() if g is not None and float(g.norm()) > 1e-9)
    qs = [rel[int(q*(len(rel)-1))] for q in (0.1, 0.5, 0.75, 0.9, 0.95, 1.0)]
    print(" grad relL2 quantiles", [(round(a,4), n[-50:]) for a, n in qs])
