set -x
mkdir -p gpurun_out/r04v
timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_token_linear_gpu.py tests/test_shadow.py -m gpu -x -q > gpurun_out/r04v/tests.log 2>&1; tail -5 gpurun_out/r04v/tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04v/bench.json 2> gpurun_out/r04v/bench.err; cat gpurun_out/r04v/bench.json | cut -c1-300
