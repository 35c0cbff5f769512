#!/bin/bash
# Kernel-time profile of the eager training step (rocprofv3 --kernel-trace --stats), summed per kernel family:
#   bash tools/profile_step.sh <tag> [bench.py flags]   ->  gpurun_out/<tag>_bench_eager_kernel_stats.csv, _by_family.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
T=${1:-prof}; shift || true
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
# (21 = the first eager step + 3 warm-up + 10 timed + the 4 drained-queue steps bench.py times the host on + its 3 profile steps; the gather's
# other-state launches are switched off so that every launch in the trace belongs to a training step)
(cd /tmp && TRANSOAR_BENCH_SKIP_OTHER=1 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/${T}_prof" -o p -- python "$OLDPWD/bench.py" --no-graph --steps 10 --warmup 3 --no-cpu-baseline "$@" > "$OUT/${T}_bench_eager_profiled.json" 2> /dev/null)
rm -f "$OUT/${T}_prof/p_kernel_trace.csv"
cp "$OUT/${T}_prof/p_kernel_stats.csv" "$OUT/${T}_bench_eager_kernel_stats.csv"
python tools/stats_by_family.py "$OUT/${T}_prof/p_kernel_stats.csv" 21 40 > "$OUT/${T}_bench_eager_by_family.txt"
head -50 "$OUT/${T}_bench_eager_by_family.txt" | cut -c1-170
