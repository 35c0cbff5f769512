set -x
mkdir -p gpurun_out/r04f
export TMPDIR=/tmp
bash tools/pmc_any.sh gpurun_out/r04f/pmc_attn roi_attn -- python tools/bench_roi_attn.py > gpurun_out/r04f/pmc_attn.txt 2>&1
rm -rf gpurun_out/r04f/pmc_attn/pass*/
tail -8 gpurun_out/r04f/pmc_attn.txt
cat > /tmp/g.py <<'PY'
import torch, sys
sys.path.insert(0, "/root/repo")
from transoar_amd import gemm
for m, k, n in ((234000, 384, 1024), (234000, 1024, 384)):
    x = torch.randn(m, k, device="cuda").bfloat16(); w = (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16(); b = torch.randn(n, device="cuda")
    for _ in range(5): gemm.linear_nt(x, w, b)
    gemm.STREAM = False
    for _ in range(5): gemm.linear_nt(x, w, b)
    gemm.STREAM = True
torch.cuda.synchronize()
PY
bash tools/pmc_any.sh gpurun_out/r04f/pmc_gemm gemm_ -- python /tmp/g.py > gpurun_out/r04f/pmc_gemm.txt 2>&1
rm -rf gpurun_out/r04f/pmc_gemm/pass*/
tail -8 gpurun_out/r04f/pmc_gemm.txt
