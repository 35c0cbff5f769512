#!/usr/bin/env python
"""Idle time between the kernels of the replayed training step, from a rocprofv3 --kernel-trace CSV of
`bench.py --steps K` (graph mode):  python tools/graph_gaps.py <p_kernel_trace.csv> [steps]
Takes the last `steps` x (kernels per step) dispatches, prints busy time, idle time and the gap histogram per step."""
import csv
import sys

path, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the timed region: find the period of the repeating sequence by the AdamW kernel (one per step)
marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
marks = marks[-(steps + 1):]
tot_busy = tot_idle = 0.0
hist = {"<1us": 0, "1-2us": 0, "2-5us": 0, "5-20us": 0, ">20us": 0}
idle_by = {k: 0.0 for k in hist}
n_k = 0
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a + 1: b + 1]
    n_k += len(seg)
    end = rows[a][1]
    for s, e, _ in seg:
        gap = max(0, s - end) / 1e3
        key = "<1us" if gap < 1 else "1-2us" if gap < 2 else "2-5us" if gap < 5 else "5-20us" if gap < 20 else ">20us"
        hist[key] += 1
        idle_by[key] += gap
        tot_idle += gap
        tot_busy += (e - max(s, end)) / 1e3 if e > end else 0.0
        end = max(end, e)
n = len(marks) - 1
print("steps", n, "kernels/step", n_k / n, "busy ms/step %.3f" % (tot_busy / n / 1e3), "idle ms/step %.3f" % (tot_idle / n / 1e3))
print("gaps per step:", {k: round(v / n, 1) for k, v in hist.items()})
print("idle ms per step by gap size:", {k: round(v / n / 1e3, 3) for k, v in idle_by.items()})
