#!/bin/bash
# LDS / issue counters per kernel for any command:  bash tools/pmc_any.sh <out_dir> <kernel-name substring filter> -- <command...>
set -u
OUT=$1; FILT=$2; shift 3
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $SET -f csv -d "$OUT/pass$i" -o p -- "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
  find "$OUT/pass$i" -name '*kernel_trace.csv' -delete
done
python - "$OUT" "$FILT" <<'PY'
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "pass*", "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(float); names = {}
    for r in csv.DictReader(open(path)):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for (d, c), v in per.items():
        n = names[d]
        if sys.argv[2] in n:
            acc[n.split("(")[0][-60:]][c].append(v)
for k, cs in sorted(acc.items()):
    w = sum(cs["SQ_WAVES"]) / len(cs["SQ_WAVES"]) if "SQ_WAVES" in cs else 1
    print(k, "waves", int(w), {c.replace("SQ_", ""): round(sum(v) / len(v) / w, 1) for c, v in sorted(cs.items()) if c != "SQ_WAVES"})
PY
