set -x
mkdir -p gpurun_out/r04a
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_msda_gpu.py -m gpu -x -q > gpurun_out/r04a/test_msda.log 2>&1; echo "rc=$?" >> gpurun_out/r04a/test_msda.log
tail -5 gpurun_out/r04a/test_msda.log
timeout 900 python -m pytest tests/test_conv3d_gpu.py tests/test_convgemm_gpu.py -m gpu -x -q > gpurun_out/r04a/test_conv.log 2>&1; echo "rc=$?" >> gpurun_out/r04a/test_conv.log
tail -5 gpurun_out/r04a/test_conv.log
timeout 600 python tools/bench_msda.py --iters 20 --dtypes bf16 > gpurun_out/r04a/msda_op_bench.jsonl 2>&1
cat gpurun_out/r04a/msda_op_bench.jsonl
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /root/repo/gpurun_out/r04a/prof -o p -- python /root/repo/tools/bench_msda.py --iters 5 --dtypes bf16 --dists model > /root/repo/gpurun_out/r04a/prof.log 2>&1
cd /root/repo; find gpurun_out/r04a/prof -name '*kernel_trace.csv' -delete
find gpurun_out/r04a/prof -name '*kernel_stats.csv' | head -1 | xargs head -20
