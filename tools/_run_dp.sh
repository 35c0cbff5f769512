set -x
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-graph --steps 30 --warmup 5 2>/dev/null | cut -c1-200
TRANSOAR_FORCE_DP=1 timeout 600 python bench.py --no-cpu-baseline --no-graph --steps 30 --warmup 5 2>/dev/null | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline --graph --steps 30 --warmup 5 2>/dev/null | cut -c1-200
TRANSOAR_FORCE_DP=1 timeout 600 python bench.py --no-cpu-baseline --graph --steps 30 --warmup 5 2>/dev/null | cut -c1-200
done
