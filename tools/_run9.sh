set -x
mkdir -p gpurun_out/r04i
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_tokens_gpu.py tests/test_gemm_gpu.py tests/test_win_attn_gpu.py "tests/test_model_parity.py::test_g9_full_width_swin_stage_on_the_kernels" tests/test_swin_gpu.py -m gpu -x -q > gpurun_out/r04i/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r04i/tests.log
grep -n "passed\|failed\|^E " gpurun_out/r04i/tests.log | head -20
timeout 900 python bench.py --swin --no-refine --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04i/bench_swin.json 2> gpurun_out/r04i/bench_swin.err; head -c 330 gpurun_out/r04i/bench_swin.json; echo; tail -2 gpurun_out/r04i/bench_swin.err | cut -c1-200
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /root/repo/gpurun_out/r04i/prof_swin -o p -- python /root/repo/bench.py --swin --no-refine --no-graph --no-cpu-baseline --steps 4 --warmup 2 > /root/repo/gpurun_out/r04i/prof_swin.log 2>&1
cd /root/repo; find gpurun_out/r04i/prof_swin -name '*kernel_trace.csv' -delete
