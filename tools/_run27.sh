set -x
mkdir -p gpurun_out/r04z
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_msda_gpu.py -m gpu -x -q -k "backward_proj or head_gather or full_size or deterministic" > gpurun_out/r04z/tests.log 2>&1; tail -3 gpurun_out/r04z/tests.log
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r04z/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --no-graph --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1)
rm -f gpurun_out/r04z/prof/p_kernel_trace.csv
python tools/stats_by_family.py gpurun_out/r04z/prof/p_kernel_stats.csv 14 60 > gpurun_out/r04z/by_family.txt; grep -i "msda3d_bwd_query\|msda3d_fwd\|kernel time" gpurun_out/r04z/by_family.txt | cut -c1-130
for i in 1 2; do
TRANSOAR_HEAD_GATHER=0 timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-200
TRANSOAR_HEAD_GATHER=1 timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-200
done
