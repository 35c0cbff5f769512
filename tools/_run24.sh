set -x
mkdir -p gpurun_out/r04x
python bench.py --swin --no-cpu-baseline > gpurun_out/r04_bench_swin.json 2> /dev/null; python tools/_pr.py gpurun_out/r04_bench_swin.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r04x/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --swin --no-graph --steps 8 --warmup 3 --no-cpu-baseline > /dev/null 2>&1)
rm -f gpurun_out/r04x/prof/p_kernel_trace.csv
python tools/stats_by_family.py gpurun_out/r04x/prof/p_kernel_stats.csv 11 25 > gpurun_out/r04_bench_swin_by_family.txt; head -36 gpurun_out/r04_bench_swin_by_family.txt | cut -c1-200
cp gpurun_out/r04x/prof/p_kernel_stats.csv gpurun_out/r04_bench_swin_eager_kernel_stats.csv
