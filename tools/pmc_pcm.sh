#!/bin/bash
# Hardware counters of the point-column forward gather (flagship shape, bf16, model-like locations; fused entry too).
#   bash tools/pmc_pcm.sh <out_dir>
set -u
OUT=${1:-gpurun_out/pmc_pcm}
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python tools/check_pcm.py --iters 2 --dists ${DISTS:-model}"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"; do
  i=$((i + 1))
  if [ -n "${PASSES:-}" ] && ! echo " $PASSES " | grep -q " $i "; then continue; fi
  rocprofv3 --kernel-trace --pmc $SET -f csv -d "$OUT/pass$i" -o p -- $CMD > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
  find "$OUT/pass$i" -name '*kernel_trace.csv' -delete
done
python tools/pmc_summary.py "$OUT" > "$OUT/summary.json"
python - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["kernels"]
for k in ("fwd_pcm", "fwd_mma"):
    if k in d:
        w = d[k].get("SQ_WAVES", 1)
        print(k, {c: round(v / w, 1) for c, v in d[k].items() if c.startswith("SQ_")}, {c: v for c, v in d[k].items() if not c.startswith("SQ_")})
PY
