set -x
mkdir -p gpurun_out/r04e
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_token_linear_gpu.py -m gpu -x -q > gpurun_out/r04e/test_gemm.log 2>&1; echo "rc=$?" >> gpurun_out/r04e/test_gemm.log
grep -n "passed\|failed\|^E " gpurun_out/r04e/test_gemm.log | head -20
timeout 600 python tools/bench_gemm.py > gpurun_out/r04e/gemm_bench.jsonl 2>&1; cat gpurun_out/r04e/gemm_bench.jsonl
timeout 1200 python -m pytest tests/test_msda_gpu.py -m gpu -x -q > gpurun_out/r04e/test_msda.log 2>&1; echo "rc=$?" >> gpurun_out/r04e/test_msda.log
tail -3 gpurun_out/r04e/test_msda.log
timeout 600 python tools/bench_msda.py --iters 20 --dtypes bf16 > gpurun_out/r04e/msda_op_bench.jsonl 2>&1; cat gpurun_out/r04e/msda_op_bench.jsonl
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /root/repo/gpurun_out/r04e/prof_msda -o p -- python /root/repo/tools/bench_msda.py --iters 5 --dtypes bf16 --dists model > /root/repo/gpurun_out/r04e/prof_msda.log 2>&1
cd /root/repo; find gpurun_out/r04e/prof_msda -name '*kernel_trace.csv' -delete
find gpurun_out/r04e/prof_msda -name '*kernel_stats.csv' | head -1 | xargs head -12 | cut -c1-200
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04e/bench_default.json 2> gpurun_out/r04e/bench_default.err; head -c 400 gpurun_out/r04e/bench_default.json
