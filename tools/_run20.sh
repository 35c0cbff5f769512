set -x
mkdir -p gpurun_out/r04t
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04t/bench.json 2> gpurun_out/r04t/bench.err; cat gpurun_out/r04t/bench.json | cut -c1-300
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r04t/tests.log 2>&1; tail -5 gpurun_out/r04t/tests.log
