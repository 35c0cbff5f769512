mkdir -p gpurun_out
timeout 600 python tools/check_q16.py --iters 10 > gpurun_out/q16_check.log 2>&1; echo "rc=$?" >> gpurun_out/q16_check.log
for upw in 1 2 4 8; do TRANSOAR_MSDA3D_Q16_UPW=$upw timeout 200 python tools/check_q16.py --time-only --dists model,init --iters 20 >> gpurun_out/q16_upw.log 2>&1; done
for pb in 1 2; do TRANSOAR_MSDA3D_Q16_PROBE=$pb timeout 200 python tools/check_q16.py --time-only --dists model,init --iters 20 >> gpurun_out/q16_probe.log 2>&1; done
cat gpurun_out/q16_check.log gpurun_out/q16_upw.log gpurun_out/q16_probe.log
