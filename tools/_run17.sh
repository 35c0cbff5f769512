set -x
mkdir -p gpurun_out/r04q
timeout 900 python -m pytest tests/test_msda_gpu.py -m gpu -x -q -k "deterministic" > gpurun_out/r04q/tests.log 2>&1; tail -25 gpurun_out/r04q/tests.log
