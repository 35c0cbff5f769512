set -x
mkdir -p gpurun_out/r04m
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q -k "wgrad384 or token_linear" > gpurun_out/r04m/tests.log 2>&1; tail -5 gpurun_out/r04m/tests.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /root/repo/gpurun_out/r04m/prof -o p -- python /root/repo/tools/bench_gemm.py > /root/repo/gpurun_out/r04m/gemm.jsonl 2>&1
cd /root/repo; find gpurun_out/r04m/prof -name '*kernel_trace.csv' -delete
grep wgrad gpurun_out/r04m/gemm.jsonl
grep -i "wgrad" gpurun_out/r04m/prof/p_kernel_stats.csv | cut -c1-200
