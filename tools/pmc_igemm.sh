#!/bin/bash
# Hardware counters of the implicit-GEMM convolution forward (csrc/conv_gemm.hip) on the FPN P2 output layer.
set -u
OUT=${1:-gpurun_out/pmc_igemm}
mkdir -p "$OUT"
export TMPDIR=/tmp
cat > /tmp/wg.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from transoar_amd import conv_gemm as G
ci, co, d, h, w = 96, 384, 40, 40, 64
x = torch.randn(2, ci, d, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
wk = G.pack_fwd(torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05)
for _ in range(3):
    G.conv_forward(x, wk, None, 1)
torch.cuda.synchronize()
PY
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $SET -f csv -d "$OUT/pass$i" -o p -- python /tmp/wg.py > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
  find "$OUT/pass$i" -name '*kernel_trace.csv' -delete
done
python - "$OUT" <<'PY'
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "pass*", "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(float); names = {}
    for r in csv.DictReader(open(path)):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for (d, c), v in per.items():
        if "igemm_kernel" in names[d]:
            acc["igemm"][c].append(v)
for k, cs in acc.items():
    w = sum(cs["SQ_WAVES"]) / len(cs["SQ_WAVES"]) if "SQ_WAVES" in cs else 1
    print(k, "waves", w, {c: round(sum(v) / len(v) / w, 1) for c, v in sorted(cs.items())})
PY
