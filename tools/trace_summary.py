#!/usr/bin/env python
"""Summarise the steady-state tail of a rocprofv3 --kernel-trace CSV:
    python tools/trace_summary.py <kernel_trace.csv> <window_ms> [top]
Groups kernels started in the last <window_ms> of the trace (MIOpen's
find/fallback kernels of the warm-up steps pollute --stats)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
win_ns = float(sys.argv[2]) * 1e6
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
end = max(int(r["End_Timestamp"]) for r in rows)
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    if int(r["Start_Timestamp"]) > end - win_ns:
        k = r["Kernel_Name"][:100]
        agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
print("busy %.2f ms in the last %.0f ms window, %d kernel names" % (tot / 1e6, win_ns / 1e6, len(agg)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print("%-100s %5d %9.3f ms %5.1f%%" % (k, v[1], v[0] / 1e6, 100.0 * v[0] / tot))
