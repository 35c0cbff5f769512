set -x
OUT=gpurun_out/r04n; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  for SH in "234000 1024 384" "234000 384 384"; do
    tag=$(echo $SH | tr ' ' _)
    timeout 300 rocprofv3 --kernel-trace --pmc $SET -f csv -d $OUT/p${i}_$tag -o p -- python tools/bench_wgrad384.py $SH > $OUT/p${i}_$tag.log 2>&1
    find $OUT/p${i}_$tag -name '*kernel_trace.csv' -delete
  done
done
python - <<'PY'
import csv,glob,collections
for tag in ("234000_1024_384","234000_384_384"):
    acc=collections.defaultdict(list)
    for path in glob.glob("gpurun_out/r04n/p*_%s/**/*counter_collection.csv"%tag, recursive=True):
        per=collections.defaultdict(float)
        for r in csv.DictReader(open(path)):
            if "wgrad384_kernel" in r["Kernel_Name"]:
                per[(r["Dispatch_Id"], r["Counter_Name"])]+=float(r["Counter_Value"])
        for (d,c),v in per.items(): acc[c].append(v)
    print(tag, {c: round(sum(v)/len(v),1) for c,v in sorted(acc.items())})
PY
