set -x
mkdir -p gpurun_out/r04fin
bash tools/_run_round.sh r04 > gpurun_out/r04fin/round.log 2>&1
grep -v "^+" gpurun_out/r04fin/round.log | grep "ms_per_step\|kernel time\|own:\|aten\|hipBLASLt\|rocclr" | cut -c1-200
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r04fin/tests.log 2>&1; tail -3 gpurun_out/r04fin/tests.log
