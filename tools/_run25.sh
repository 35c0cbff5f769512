set -x
mkdir -p gpurun_out/r04y
timeout 1200 python -m pytest tests/test_msda_gpu.py -m gpu -x -q -k "backward_proj or head_gather or deterministic" > gpurun_out/r04y/tests.log 2>&1; tail -15 gpurun_out/r04y/tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04y/bench.json 2> gpurun_out/r04y/bench.err; cat gpurun_out/r04y/bench.json | cut -c1-300
