set -x
mkdir -p gpurun_out/r04c
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_win_attn_gpu.py "tests/test_model_parity.py::test_g9_full_width_swin_stage_on_the_kernels" -m gpu -x -q > gpurun_out/r04c/test_win_attn.log 2>&1; echo "rc=$?" >> gpurun_out/r04c/test_win_attn.log
tail -30 gpurun_out/r04c/test_win_attn.log
timeout 900 python -m pytest tests/test_swin_gpu.py -m gpu -x -q > gpurun_out/r04c/test_swin.log 2>&1; echo "rc=$?" >> gpurun_out/r04c/test_swin.log
tail -8 gpurun_out/r04c/test_swin.log
timeout 900 python bench.py --swin --no-refine --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04c/bench_swin.json 2> gpurun_out/r04c/bench_swin.err; head -c 700 gpurun_out/r04c/bench_swin.json; tail -3 gpurun_out/r04c/bench_swin.err
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /root/repo/gpurun_out/r04c/prof_swin -o p -- python /root/repo/bench.py --swin --no-refine --no-graph --no-cpu-baseline --steps 4 --warmup 2 > /root/repo/gpurun_out/r04c/prof_swin.log 2>&1
cd /root/repo; find gpurun_out/r04c/prof_swin -name '*kernel_trace.csv' -delete
bash tools/collect_msda_pmc.sh gpurun_out/r04c/msda_pmc > gpurun_out/r04c/msda_pmc.log 2>&1
rm -rf gpurun_out/r04c/msda_pmc/pass*/ 2>/dev/null; ls gpurun_out/r04c/msda_pmc | head; head -c 3000 gpurun_out/r04c/msda_pmc/summary.json
