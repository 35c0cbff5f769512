set -x
mkdir -p gpurun_out/r04w
timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_token_linear_gpu.py -m gpu -x -q > gpurun_out/r04w/tests.log 2>&1; tail -3 gpurun_out/r04w/tests.log
timeout 300 python tools/bench_gemm.py 2>/dev/null | grep -v wgrad | cut -c1-200
TRANSOAR_GEMM_PERSIST_WGS=0 timeout 300 python tools/bench_gemm.py 2>/dev/null | grep -v wgrad | cut -c1-120
TRANSOAR_GEMM_PERSIST_WGS=768 timeout 300 python tools/bench_gemm.py 2>/dev/null | grep -v wgrad | cut -c1-120
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04w/bench.json 2> gpurun_out/r04w/bench.err; cat gpurun_out/r04w/bench.json | cut -c1-300
