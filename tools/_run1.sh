mkdir -p gpurun_out
rm -f gpurun_out/q16_*.log
timeout 600 python tools/check_q16.py --iters 10 > gpurun_out/q16_check.log 2>&1; echo "rc=$?" >> gpurun_out/q16_check.log
for pb in 1 2; do TRANSOAR_MSDA3D_Q16_PROBE=$pb timeout 200 python tools/check_q16.py --time-only --dists model,init --iters 20 >> gpurun_out/q16_probe.log 2>&1; done
grep -v amdgpu.ids gpurun_out/q16_check.log gpurun_out/q16_probe.log
TRANSOAR_MSDA3D_Q16_PROBE=0 bash tools/pmc_any.sh gpurun_out/pmc_q16_p0 q16 -- python tools/check_q16.py --time-only --dists model --iters 5 > gpurun_out/pmc_q16_p0.txt 2>&1
cat gpurun_out/pmc_q16_p0.txt; rm -rf gpurun_out/pmc_q16_p0
