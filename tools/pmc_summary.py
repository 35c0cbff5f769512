#!/usr/bin/env python
"""rocprofv3 --pmc counter CSVs (tools/collect_msda_pmc.sh) -> per-kernel means as JSON.
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB and
FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM)."""
import collections, csv, glob, json, os, re, sys

root = sys.argv[1]
SHORT = [("msda3d_fwd_pcm", "fwd_pcm"), ("msda3d_fwd_mma", "fwd_mma"), ("msda3d_bwd_query_mma", "bwd_query_mma"), ("msda3d_fwd_brick", "fwd_brick"), ("msda3d_bwd_query_brick", "bwd_query_brick"),
         ("msda3d_cell_fill_w8", "cell_fill_w8"), ("msda3d_bwd_value_tile", "bwd_value_tile"),
         ("msda3d_bwd_value_cells", "bwd_value_cells"), ("msda3d_coarse_rows_store", "coarse_rows_store"),
         ("msda3d_scan_tiles", "scan_tiles"), ("msda3d_fwd_vec", "fwd_vec"), ("msda3d_bwd_query_vec", "bwd_query_vec"),
         ("msda3d_bwd_value_pull", "value_pull"), ("msda3d_cell_fill", "cell_fill"), ("msda3d_cell_count", "cell_count"),
         ("msda3d_scan_add", "scan_add"), ("msda3d_scan_tile_sums", "scan_tile_sums"), ("msda3d_zero16", "zero16")]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(path)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per_dispatch[key] += float(r["Counter_Value"])          # rows are per dimension (XCD/SE): sum
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for (disp, counter), v in per_dispatch.items():
        for pat, short in SHORT:
            if pat in names[disp]:
                acc[short][counter].append(v)
                break
out = {}
for k, counters in acc.items():
    out[k] = {c: sum(v) / len(v) for c, v in sorted(counters.items())}
    out[k]["launches_sampled"] = max(len(v) for v in counters.values())
    if "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]:
        out[k]["hbm_bytes_per_launch"] = (2.0 * out[k]["FETCH_SIZE"] + out[k]["WRITE_SIZE"]) * 1024.0
print(json.dumps({
    "command": "bash tools/collect_msda_pmc.sh  (rocprofv3 --kernel-trace --pmc <set> -- %s)" % (sys.argv[2] if len(sys.argv) > 2 else "python tools/bench_msda.py --iters 2 --dists model --dtypes bf16"),
    "shape": {"N": 2, "S": 117000, "M": 6, "C": 64, "L": 4, "Lq": 117000, "P": 4, "value_dtype": "bf16", "loc_dtype": "f32"},
    "note": "mean over the launches of one run (warm-up launches included); one rocprofv3 pass per counter set; "
            "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KiB, the factor 2 being the guide's gfx950 correction "
            "for wide coalesced reads; calibrated on 128-byte row gathers with known traffic as well "
            "(profiles/r02_fetch_calibration.txt: known / counter = 2.0)",
    "kernels": out}, indent=1))
