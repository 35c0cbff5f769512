import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transoar_amd import gemm, conv_gemm, token_linear as tl, tokens
g = torch.Generator(device="cuda").manual_seed(5)
m = 50001
x = torch.randn(m, 384, device="cuda", generator=g).to(torch.bfloat16)
cap = {}
orig = gemm.wgrad384
def wrap(gy, xx):
    out = orig(gy, xx)
    cap["gy"], cap["x"], cap["out"] = gy.clone(), xx.clone(), out.clone()
    return out
gemm.wgrad384 = wrap
lin = torch.nn.Linear(384, 1024).cuda()
drop = torch.nn.Dropout(0.1).train()
xg = x.clone().requires_grad_()
with torch.autocast("cuda", dtype=torch.bfloat16):
    yy = tl.linear_relu_dropout(xg, lin.weight, lin.bias, drop)
gy = torch.randn(yy.shape, device="cuda", generator=g).to(torch.bfloat16)
yy.backward(gy)
ref = (cap["gy"].double().t() @ cap["x"].double()).float()
print("captured shapes", cap["gy"].shape, cap["x"].shape, cap["gy"].is_contiguous(), cap["x"].is_contiguous())
err = (cap["out"] - ref).abs()
print("in-graph err", err.max().item(), "ref max", ref.abs().max().item())
again = orig(cap["gy"], cap["x"])
print("again err", (again - ref).abs().max().item(), "old", (conv_gemm.linear_wgrad(cap["x"], cap["gy"]) - ref).abs().max().item())
bad = (err > 1e-2).nonzero()
print("bad", bad.shape[0])
if bad.shape[0]:
    print(torch.unique(bad[:, 0])[:50].tolist()); print(torch.unique(bad[:, 1])[:50].tolist())
print("nan in gy", torch.isnan(cap["gy"].float()).any().item(), "frac zero", (cap["gy"] == 0).float().mean().item())
# decomposition of the test's weight-gradient check
for trial in range(3):
    lin = torch.nn.Linear(384, 1024).cuda()
    xg = x.clone().requires_grad_()
    torch.manual_seed(11)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yy = tl.linear_relu_dropout(xg, lin.weight, lin.bias, drop)
    yy.backward(gy)
    torch.manual_seed(11)
    seed2 = tokens.dropout_seed(x)
    keep2 = tokens.hashed_keep(seed2, m * 1024, 0.9).view(m, 1024).float()
    xr = x.float().requires_grad_()
    wr = lin.weight.detach().to(torch.bfloat16).float().requires_grad_()
    pre = torch.nn.functional.linear(xr, wr, lin.bias.detach())
    yr = torch.relu(pre) * keep2 / 0.9
    yr.backward(gy.float())
    rel = lambda a, c: float((a.float() - c).abs().max() / c.abs().max())
    ref_b = (cap["gy"].double().t() @ cap["x"].double()).float()
    mism = ((yy.float() > 0) != (yr > 0)).float().sum().item()
    print(trial, "rel(gw, fp32 ref)", rel(lin.weight.grad, wr.grad), "rel(gw, exact of captured gh)", rel(lin.weight.grad, ref_b),
          "mask mismatches", mism, "rel(y)", rel(yy, yr))
