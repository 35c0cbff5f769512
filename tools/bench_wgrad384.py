#!/usr/bin/env python
"""The token-streaming weight gradient alone (for counter passes): tools/bench_wgrad384.py [T N K] [iters]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transoar_amd import gemm  # noqa: E402

t, n, k = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (234000, 1024, 384)
iters = int(sys.argv[4]) if len(sys.argv) >= 5 else 5
gy = torch.randn(t, n, device="cuda").bfloat16()
x = torch.randn(t, k, device="cuda").bfloat16()
for _ in range(iters):
    gemm.wgrad384(gy, x)
torch.cuda.synchronize()
