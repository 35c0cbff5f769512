set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; tail -c 3000 gpurun_out/r02_bench_default.json
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof_bench -o p -- python $GRAFT_REPO_ROOT/bench.py --no-graph --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02_bench_profiled.json 2> /dev/null)
rm -f gpurun_out/r02_prof_bench/p_kernel_trace.csv
bash tools/collect_msda_pmc.sh gpurun_out/r02_msda_pmc > /dev/null 2>&1
ls gpurun_out/r02_msda_pmc | head
python -m pytest tests/test_model_parity.py tests/test_amos_gpu.py tests/test_train_step_gpu.py -q -m gpu 2>&1 | grep -v Warning | tail -12
