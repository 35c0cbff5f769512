#!/bin/bash
# Collect hardware counters for the MSDeformAttn-3D kernels on the flagship shape.
# Run on the GPU box from the repo root:   bash tools/collect_msda_pmc.sh <out_dir>
# One rocprofv3 pass per counter set (gfx950 slot limits: TCC 4, SQ 8, GRBM 2;
# FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2) -- and never together with the
# sys/hip/hsa trace domains.  tools/pmc_summary.py turns the CSVs into the JSON
# committed under profiles/.
set -u
OUT=${1:-gpurun_out/msda_pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
# CMD: what to profile (default: the op bench on jittered locations); PASSES: which counter sets (default all 6).
#   CMD="python bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline" PASSES="1 2 3" bash tools/collect_msda_pmc.sh <dir>
# collects the HBM traffic of the training step's OWN launches (profiles/r05_msda_pmc_step.json, read by bench.py).
CMD=${CMD:-python tools/bench_msda.py --iters 2 --dists model --dtypes bf16}
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"; do
  i=$((i + 1))
  if [ -n "${PASSES:-}" ] && ! echo " $PASSES " | grep -q " $i "; then continue; fi
  rocprofv3 --kernel-trace --pmc $SET -f csv -d "$OUT/pass$i" -o p -- $CMD > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
  # keep only what the summary needs (the box returns at most 64 MiB)
  find "$OUT/pass$i" -name '*kernel_trace.csv' -delete
done
python tools/pmc_summary.py "$OUT" "$CMD" > "$OUT/summary.json"
ls -la "$OUT"
