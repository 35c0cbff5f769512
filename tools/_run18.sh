set -x
mkdir -p gpurun_out/r04r
timeout 900 python -m pytest tests/test_conv3d_gpu.py -m gpu -x -q > gpurun_out/r04r/tests.log 2>&1; tail -4 gpurun_out/r04r/tests.log
MIOPEN_FIND_MODE=FAST timeout 300 python tools/bench_conv.py s0c2 > gpurun_out/r04r/conv.jsonl 2>&1; tail -2 gpurun_out/r04r/conv.jsonl | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04r/bench.json 2> gpurun_out/r04r/bench.err; cat gpurun_out/r04r/bench.json | cut -c1-300
