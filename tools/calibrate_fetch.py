#!/usr/bin/env python
"""Calibration of the FETCH_SIZE counter for the access patterns of the MSDeformAttn kernels: kernels with a KNOWN
number of bytes read -- a wide coalesced copy, and 128-byte row gathers (contiguous rows; rows at a 768-byte pitch
like one head of value (N, S, M, C)) -- run under  rocprofv3 --pmc FETCH_SIZE.  tools/calibrate_fetch.sh prints
counter / known bytes per kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transoar_amd import rows  # noqa: E402

torch.manual_seed(0)
dev = "cuda"
n_rows = 1_404_000                                   # 2 * 117000 * 6 rows of 64 bf16
x = torch.randn(1, n_rows, 64, device=dev).to(torch.bfloat16)
perm = torch.randperm(n_rows, device=dev).int()
big = torch.randn(256 << 20, device=dev).to(torch.bfloat16)          # 512 MiB
xs = torch.randn(1, n_rows // 6, 384, device=dev).to(torch.bfloat16)  # 768-byte rows; head 0 = first 128 bytes
perm_s = torch.randperm(n_rows // 6, device=dev).int()
for _ in range(3):
    y0 = big.clone()                                  # wide coalesced: reads 512 MiB
    y1 = rows.gather(x, perm)                         # 128-byte rows, random order: reads n_rows * 128 B (+ index)
    y2 = rows.gather(xs, perm_s)                      # 768-byte rows, random order
    y3 = xs[:, :, :64].contiguous()                   # 128 bytes of every 768-byte row, in order
torch.cuda.synchronize()
print("known_bytes clone %d gather128 %d gather768 %d strided128 %d" % (
    big.numel() * 2, n_rows * 128 + n_rows * 4, (n_rows // 6) * 768 + (n_rows // 6) * 4, (n_rows // 6) * 128))
