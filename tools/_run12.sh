set -x
mkdir -p gpurun_out/r04l
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q -k "wgrad384 or token_linear" > gpurun_out/r04l/tests.log 2>&1; tail -15 gpurun_out/r04l/tests.log
timeout 300 python tools/bench_gemm.py > gpurun_out/r04l/gemm.jsonl 2>&1; grep wgrad gpurun_out/r04l/gemm.jsonl
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04l/bench.json 2> gpurun_out/r04l/bench.err; cat gpurun_out/r04l/bench.json | cut -c1-300
