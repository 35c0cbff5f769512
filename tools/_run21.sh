set -x
mkdir -p gpurun_out/r04u
timeout 2400 python -m pytest tests/test_shadow.py tests/test_swin_gpu.py tests/test_token_linear_gpu.py tests/test_tokens_gpu.py tests/test_train_step_gpu.py tests/test_visceral_gpu.py tests/test_win_attn_gpu.py tests/test_model_parity.py tests/test_amos_gpu.py -m gpu -x -q > gpurun_out/r04u/tests_tail.log 2>&1; tail -5 gpurun_out/r04u/tests_tail.log
bash tools/_run_round.sh r04 > gpurun_out/r04u/round.log 2>&1
grep -v "^+" gpurun_out/r04u/round.log | tail -60
