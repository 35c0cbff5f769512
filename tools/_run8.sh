set -x
mkdir -p gpurun_out/r04h
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_roi_attn_gpu.py tests/test_gemm_gpu.py tests/test_token_linear_gpu.py tests/test_train_step_gpu.py -m gpu -x -q > gpurun_out/r04h/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r04h/tests.log
grep -n "passed\|failed\|^E " gpurun_out/r04h/tests.log | head -20
timeout 300 python tools/bench_roi_attn.py > gpurun_out/r04h/roi_attn_bench.jsonl 2>&1; grep roi_att gpurun_out/r04h/roi_attn_bench.jsonl | cut -c1-250
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04h/bench_default.json 2> gpurun_out/r04h/bench_default.err; head -c 330 gpurun_out/r04h/bench_default.json; echo
TRANSOAR_FORCE_DP=1 timeout 900 python bench.py --graph --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04h/bench_one_rank_rccl_graph.json 2> gpurun_out/r04h/bench_one_rank_rccl_graph.err; head -c 330 gpurun_out/r04h/bench_one_rank_rccl_graph.json; echo; grep -o '"step_mode": "[^"]*"' gpurun_out/r04h/bench_one_rank_rccl_graph.json
TRANSOAR_FORCE_DP=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04h/bench_one_rank_rccl.json 2> gpurun_out/r04h/bench_one_rank_rccl.err; head -c 330 gpurun_out/r04h/bench_one_rank_rccl.json; echo
timeout 900 python bench.py --no-graph --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04h/bench_eager.json 2> /dev/null; head -c 330 gpurun_out/r04h/bench_eager.json; echo
