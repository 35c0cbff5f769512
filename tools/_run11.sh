set -x
mkdir -p gpurun_out/r04k
timeout 900 python -m pytest tests/test_msda_gpu.py tests/test_token_linear_gpu.py tests/test_gemm_gpu.py tests/test_model_parity.py -m gpu -x -q > gpurun_out/r04k/tests.log 2>&1; tail -3 gpurun_out/r04k/tests.log
timeout 300 python tools/bench_msda.py > gpurun_out/r04k/msda_op.jsonl 2>&1; tail -2 gpurun_out/r04k/msda_op.jsonl
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04k/bench.json 2> gpurun_out/r04k/bench.err; cat gpurun_out/r04k/bench.json | cut -c1-400
