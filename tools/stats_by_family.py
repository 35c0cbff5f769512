#!/usr/bin/env python
"""rocprofv3 --stats kernel_stats.csv of `bench.py --no-graph --steps K --warmup W` -> ms per step by kernel family and
the top kernels.   python tools/stats_by_family.py <kernel_stats.csv> <steps incl. warm-up and the first eager step>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
fam = {}


def family(n):
    if n.startswith("void ck::") or "ck16tensor_operation" in n or "miopen" in n.lower() or "batched_transpose" in n:
        return "MIOpen/CK"
    if n.startswith("Cijk_"):
        return "hipBLASLt"
    if "at::native" in n or "at_cuda_detail" in n:
        return "aten"
    if "rocclr" in n or n in ("attn_fwd",) or n.startswith("bwd_kernel"):
        return "rocclr/aotriton"
    if "msda3d" in n:
        return "own: msda3d"
    if "conv3d" in n or "layout_" in n:
        return "own: conv"
    if "instnorm" in n:
        return "own: instnorm"
    if "gemm_nt" in n:
        return "own: gemm"
    return "own: tokens/rows/other"


for r in rows:
    f = family(r["Name"])
    a = fam.setdefault(f, [0.0, 0])
    a[0] += float(r["TotalDurationNs"]) / 1e6 / steps
    a[1] += int(r["Calls"]) / steps
tot = sum(v[0] for v in fam.values())
print("kernel time per step %.2f ms" % tot)
for f, (ms, calls) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print("%-28s %7.2f ms %8.1f launches" % (f, ms, calls))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%8.3f ms/step %7.1f calls avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps,
                                                        float(r["AverageNs"]) / 1e3, r["Name"][:120]))
