set -x
mkdir -p gpurun_out/r04o
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_token_linear_gpu.py -m gpu -x -q > gpurun_out/r04o/tests.log 2>&1; tail -3 gpurun_out/r04o/tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04o/bench.json 2> gpurun_out/r04o/bench.err; cat gpurun_out/r04o/bench.json | cut -c1-300
