# The round's measurement sequence on the GPU box (from the repo root):  bash tools/_run_round.sh [tag]
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r04}
python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; python tools/_pr.py gpurun_out/${T}_bench_default.json
python bench.py --no-graph --no-cpu-baseline > gpurun_out/${T}_bench_eager.json 2> /dev/null; python tools/_pr.py gpurun_out/${T}_bench_eager.json
python bench.py --no-refine --no-cpu-baseline > gpurun_out/${T}_bench_no_refine.json 2> /dev/null; python tools/_pr.py gpurun_out/${T}_bench_no_refine.json
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/${T}_prof_bench -o p -- python $GRAFT_REPO_ROOT/bench.py --no-graph --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${T}_bench_eager_profiled.json 2> /dev/null)
rm -f gpurun_out/${T}_prof_bench/p_kernel_trace.csv
python tools/stats_by_family.py gpurun_out/${T}_prof_bench/p_kernel_stats.csv 14 12 > gpurun_out/${T}_bench_eager_by_family.txt; head -12 gpurun_out/${T}_bench_eager_by_family.txt
bash tools/collect_msda_pmc.sh gpurun_out/${T}_msda_pmc > /dev/null 2>&1
python tools/bench_msda.py --iters 20 --dtypes bf16 > gpurun_out/${T}_msda_op_bench.jsonl 2>/dev/null; cut -c1-200 gpurun_out/${T}_msda_op_bench.jsonl
python tools/check_pcm.py --dists model,init,uniform,wide > gpurun_out/${T}_msda_fwd_kernels.jsonl 2>/dev/null; tail -3 gpurun_out/${T}_msda_fwd_kernels.jsonl
python tools/bench_gemm.py > gpurun_out/${T}_gemm_bench.jsonl 2>/dev/null
python tools/bench_convgemm.py > gpurun_out/${T}_conv_layers.jsonl 2>/dev/null; cut -c1-220 gpurun_out/${T}_conv_layers.jsonl | head -4
TRANSOAR_FORCE_DP=1 python bench.py --no-cpu-baseline --no-graph --steps 20 --warmup 5 > gpurun_out/${T}_bench_one_rank_rccl.json 2>/dev/null; python tools/_pr.py gpurun_out/${T}_bench_one_rank_rccl.json
TRANSOAR_FORCE_DP=1 python bench.py --no-cpu-baseline --graph --steps 20 --warmup 5 > gpurun_out/${T}_bench_one_rank_rccl_graph.json 2>/dev/null; python tools/_pr.py gpurun_out/${T}_bench_one_rank_rccl_graph.json
python bench.py --swin --no-refine --no-cpu-baseline > gpurun_out/${T}_bench_swin.json 2> /dev/null; python tools/_pr.py gpurun_out/${T}_bench_swin.json
python bench.py --swin --no-cpu-baseline > gpurun_out/${T}_bench_swin_refine.json 2> /dev/null; python tools/_pr.py gpurun_out/${T}_bench_swin_refine.json
python tools/bench_roi_attn.py > gpurun_out/${T}_roi_attn_bench.jsonl 2>/dev/null; tail -3 gpurun_out/${T}_roi_attn_bench.jsonl | cut -c1-220
TRANSOAR_MSDA_DETERMINISTIC=1 python tools/bench_msda.py --iters 10 --dtypes bf16 --dists model > gpurun_out/${T}_msda_op_bench_deterministic.jsonl 2>/dev/null; cut -c1-200 gpurun_out/${T}_msda_op_bench_deterministic.jsonl
