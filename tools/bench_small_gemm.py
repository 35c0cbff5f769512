#!/usr/bin/env python
"""The Focused Decoder's small products (M = 1 080 ... 3 240 rows) on hipBLASLt (F.linear) and on the tiled kernel of
csrc/gemm.hip: they are launch-latency bound at 12-16 us in the library.    python tools/bench_small_gemm.py"""
import json
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from transoar_amd import gemm  # noqa: E402


def timeit(fn, n=100):
    """GPU time per call: the calls are captured into one HIP graph and replayed (launched eagerly these products are bound by
    the host: 14-19 us of Python + library overhead per call)."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(n):
            fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * n) * 1e3


for m, k, n in ((1080, 384, 384), (1080, 384, 768), (1080, 384, 1024), (1080, 1024, 384), (3240, 384, 384), (400, 3072, 384), (400, 384, 3072)):
    x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    b16, b32 = torch.randn(n, device="cuda", dtype=torch.bfloat16), torch.randn(n, device="cuda")
    lib_us = timeit(lambda: F.linear(x, w, b16))
    own_us = timeit(lambda: gemm.linear_nt(x, w, b32))
    err = float((gemm.linear_nt(x, w, b32).float() - F.linear(x.float(), w.float(), b32)).abs().max())
    print(json.dumps({"M": m, "K": k, "N": n, "hipblaslt_us": round(lib_us, 2), "own_tiled_us": round(own_us, 2), "max_abs_err_vs_fp32": round(err, 4)}))
