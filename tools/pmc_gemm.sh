set -u
OUT=gpurun_out/pmc_gemm
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET -f csv -d $OUT/pass$i -o p -- python tools/bench_gemm.py > $OUT/pass$i.log 2>&1
  find $OUT/pass$i -name '*kernel_trace.csv' -delete
done
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/pmc_gemm/pass*/**/*counter_collection.csv", recursive=True):
    per=collections.defaultdict(float); names={}
    for r in csv.DictReader(open(path)):
        per[(r["Dispatch_Id"], r["Counter_Name"])]+=float(r["Counter_Value"]); names[r["Dispatch_Id"]]=r["Kernel_Name"]
    for (d,c),v in per.items():
        n=names[d]
        key="ours" if "gemm_nt_kernel" in n else ("blas" if n.startswith("Cijk") else None)
        if key: acc[key][c].append(v)
for k,cs in acc.items():
    w=sum(cs["SQ_WAVES"])/len(cs["SQ_WAVES"]) if "SQ_WAVES" in cs else 1
    print(k, {c: round(sum(v)/len(v)/w,1) if c.startswith("SQ_") else round(sum(v)/len(v),1) for c,v in sorted(cs.items())})
PY
