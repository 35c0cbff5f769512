#!/usr/bin/env python
"""csrc/gemm.hip vs hipBLASLt (torch F.linear) on the refine block's token shapes, bf16, bias in the epilogue."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transoar_amd import gemm  # noqa: E402


def time_ms(fn, iters=20):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ts[len(ts) // 2]


for m, k, n in ((234000, 384, 384), (234000, 384, 1024), (234000, 1024, 384), (204800, 384, 384), (234000, 384, 576), (204800, 384, 3072)):
    x = torch.randn(m, k, device="cuda").bfloat16()
    w = (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16()
    b = torch.randn(n, device="cuda")
    bb = b.bfloat16()
    ours = time_ms(lambda: gemm.linear_nt(x, w, b))
    gemm.STREAM = False
    tiled = time_ms(lambda: gemm.linear_nt(x, w, b))            # round 2's 128 x 128 LDS-tiled kernel (csrc/gemm.hip)
    gemm.STREAM = True
    blas = time_ms(lambda: torch.nn.functional.linear(x, w, bb))
    fl = 2.0 * m * k * n
    print(json.dumps({"M": m, "K": k, "N": n, "kernel": gemm.stream_kind(x, w) or "tiled", "ours_ms": round(ours, 4),
                      "tiled_ms": round(tiled, 4), "hipblaslt_ms": round(blas, 4),
                      "ours_TFs": round(fl / ours / 1e9, 1), "hipblaslt_TFs": round(fl / blas / 1e9, 1),
                      "frac_of_2.5PF": round(fl / ours / 1e9 / 2500, 3)}), flush=True)

# weight gradients dW = dY^T X (contraction over the tokens): the one-tap case of csrc/conv_gemm.hip's voxel-major GEMM
# against the chunked hipBLASLt batch of round 2
from transoar_amd import conv_gemm, token_linear  # noqa: E402

for m, k, n in ((234000, 384, 384), (234000, 384, 1024), (234000, 1024, 384), (234000, 384, 576), (204800, 96, 96), (204800, 384, 3072)):
    x = torch.randn(m, k, device="cuda").bfloat16()
    gy = torch.randn(m, n, device="cuda").bfloat16()
    ours = time_ms(lambda: token_linear.weight_grad(gy, x))          # round 4: wgrad384 where it applies
    voxel = time_ms(lambda: conv_gemm.linear_wgrad(x, gy))           # round 3: the one-tap case of the voxel-major conv GEMM
    token_linear.USE_HIP_WGRAD = False
    blas = time_ms(lambda: token_linear.weight_grad(gy, x))
    token_linear.USE_HIP_WGRAD = True
    fl = 2.0 * m * k * n
    print(json.dumps({"wgrad": True, "T": m, "K": k, "N": n, "kernel": "wgrad384" if gemm.wgrad384_usable(gy, x) else "conv_gemm",
                      "ours_ms": round(ours, 4), "conv_gemm_ms": round(voxel, 4), "hipblaslt_chunked_ms": round(blas, 4),
                      "ours_TFs": round(fl / ours / 1e9, 1), "frac_of_2.5PF": round(fl / ours / 1e9 / 2500, 3)}), flush=True)
