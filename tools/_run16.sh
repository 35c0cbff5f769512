set -x
mkdir -p gpurun_out/r04p
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_token_linear_gpu.py tests/test_roi_attn_gpu.py tests/test_win_attn_gpu.py -m gpu -x -q > gpurun_out/r04p/tests.log 2>&1; tail -3 gpurun_out/r04p/tests.log
timeout 300 python tools/bench_gemm.py > gpurun_out/r04p/gemm.jsonl 2>&1; cut -c1-250 gpurun_out/r04p/gemm.jsonl
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04p/bench.json 2> gpurun_out/r04p/bench.err; cat gpurun_out/r04p/bench.json | cut -c1-300
