#!/bin/bash
# Hardware counters of the forward gather alone (flagship shape, bf16, model-like locations).
#   bash tools/pmc_fwd.sh <out_dir> [extra env assignments]
set -u
OUT=${1:-gpurun_out/pmc_fwd}
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python tools/check_mma.py --flagship-only --dists ${DISTS:-model}"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $SET -f csv -d "$OUT/pass$i" -o p -- $CMD > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
  find "$OUT/pass$i" -name '*kernel_trace.csv' -delete
done
python tools/pmc_summary.py "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
