set -x
mkdir -p gpurun_out/r04z
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r04z/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --no-graph --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1)
rm -f gpurun_out/r04z/prof/p_kernel_trace.csv
python tools/stats_by_family.py gpurun_out/r04z/prof/p_kernel_stats.csv 14 60 > gpurun_out/r04z/by_family.txt; grep -i "msda3d\|sampling_head\|kernel time" gpurun_out/r04z/by_family.txt | cut -c1-130
