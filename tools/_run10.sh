set -x
mkdir -p gpurun_out/r04j
export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /root/repo/gpurun_out/r04j/prof -o p -- python /root/repo/bench.py --no-graph --no-cpu-baseline --steps 10 --warmup 3 > /root/repo/gpurun_out/r04j/prof.log 2>&1
cd /root/repo; find gpurun_out/r04j/prof -name '*kernel_trace.csv' -delete
python tools/stats_by_family.py gpurun_out/r04j/prof/p_kernel_stats.csv 14 60 > gpurun_out/r04j/by_family.txt; head -14 gpurun_out/r04j/by_family.txt
