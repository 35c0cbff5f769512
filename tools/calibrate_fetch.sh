#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of kernels with known traffic (see tools/calibrate_fetch.py); run on the GPU box from the repo root
OUT=${1:-gpurun_out/calib_fetch}
mkdir -p "$OUT"; export TMPDIR=/tmp
for SET in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $SET -f csv -d "$OUT/$SET" -o p -- python tools/calibrate_fetch.py > "$OUT/$SET.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, collections, glob, sys, os
out = sys.argv[1]
for cset in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for path in glob.glob(os.path.join(out, cset, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"][:70]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        v = sorted(d.values())
        print(cset, "%-72s launches %3d  median KiB %12.1f  = %8.1f MB" % (k, len(v), v[len(v) // 2], v[len(v) // 2] * 1024 / 1e6))
print(open(os.path.join(out, "FETCH_SIZE.log")).read().strip().split("\n")[-1])
PY
