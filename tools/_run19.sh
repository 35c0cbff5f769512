set -x
cat > /tmp/one_conv.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
from transoar_amd import conv3d as C
x = torch.randn(2, 24, 160, 160, 256, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
wt = torch.randn(24, 24, 3, 3, 3, device="cuda") * 0.05
wk = C._pack_taps(wt)
for _ in range(4):
    y = C.conv3d_k3_forward(x, wk, None, 1)
torch.cuda.synchronize()
PY
bash tools/pmc_any.sh gpurun_out/r04s conv3d_k3_lds -- python /tmp/one_conv.py > gpurun_out/r04s_summary.txt 2>&1
cat gpurun_out/r04s_summary.txt | tail -5
rm -rf gpurun_out/r04s
