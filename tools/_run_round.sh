# The round's measurement sequence on the GPU box (from the repo root):  bash tools/_run_round.sh
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; python tools/_pr.py gpurun_out/r02_bench_default.json
python bench.py --no-graph --no-cpu-baseline > gpurun_out/r02_bench_eager.json 2> /dev/null; python tools/_pr.py gpurun_out/r02_bench_eager.json
python bench.py --no-refine --no-cpu-baseline > gpurun_out/r02_bench_no_refine.json 2> /dev/null; python tools/_pr.py gpurun_out/r02_bench_no_refine.json
python bench.py --swin --no-refine --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r02_bench_swin.json 2> /dev/null; python tools/_pr.py gpurun_out/r02_bench_swin.json
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof_bench -o p -- python $GRAFT_REPO_ROOT/bench.py --no-graph --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02_bench_eager_profiled.json 2> /dev/null)
rm -f gpurun_out/r02_prof_bench/p_kernel_trace.csv
bash tools/collect_msda_pmc.sh gpurun_out/r02_msda_pmc > /dev/null 2>&1
python tools/bench_msda.py --iters 20 --dtypes bf16 > gpurun_out/r02_msda_op_bench.jsonl 2>/dev/null; cat gpurun_out/r02_msda_op_bench.jsonl | cut -c1-200
python tools/bench_gemm.py > gpurun_out/r02_gemm_bench.jsonl 2>/dev/null
python tools/bench_conv.py > gpurun_out/r02_conv_layers.jsonl 2>/dev/null; cut -c1-160 gpurun_out/r02_conv_layers.jsonl | head -4
TRANSOAR_FORCE_DP=1 python bench.py --no-cpu-baseline --no-graph --steps 20 --warmup 5 > gpurun_out/r02_bench_one_rank_rccl.json 2>/dev/null; python tools/_pr.py gpurun_out/r02_bench_one_rank_rccl.json
TRANSOAR_FORCE_DP=1 python bench.py --no-cpu-baseline --graph --steps 20 --warmup 5 > gpurun_out/r02_bench_one_rank_rccl_graph.json 2>/dev/null; python tools/_pr.py gpurun_out/r02_bench_one_rank_rccl_graph.json
