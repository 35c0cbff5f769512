set -x
mkdir -p gpurun_out/r04b
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_roi_attn_gpu.py -m gpu -x -q > gpurun_out/r04b/test_roi_attn.log 2>&1; echo "rc=$?" >> gpurun_out/r04b/test_roi_attn.log
tail -25 gpurun_out/r04b/test_roi_attn.log
timeout 300 python tools/bench_roi_attn.py > gpurun_out/r04b/roi_attn_bench.jsonl 2>&1; cat gpurun_out/r04b/roi_attn_bench.jsonl
timeout 1200 python -m pytest tests/test_msda_gpu.py -m gpu -x -q > gpurun_out/r04b/test_msda.log 2>&1; echo "rc=$?" >> gpurun_out/r04b/test_msda.log
tail -4 gpurun_out/r04b/test_msda.log
timeout 600 python tools/bench_msda.py --iters 20 --dtypes bf16 --dists model > gpurun_out/r04b/msda_op_bench.jsonl 2>&1; cat gpurun_out/r04b/msda_op_bench.jsonl
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04b/bench_default.json 2> gpurun_out/r04b/bench_default.err; tail -c 1500 gpurun_out/r04b/bench_default.json; tail -5 gpurun_out/r04b/bench_default.err
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /root/repo/gpurun_out/r04b/prof_swin -o p -- python /root/repo/bench.py --swin --no-graph --no-cpu-baseline --steps 4 --warmup 2 > /root/repo/gpurun_out/r04b/prof_swin.log 2>&1
cd /root/repo; find gpurun_out/r04b/prof_swin -name '*kernel_trace.csv' -delete
find gpurun_out/r04b/prof_swin -name '*kernel_stats.csv' | head -1 | xargs head -40 | cut -c1-260
