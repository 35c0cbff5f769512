#!/bin/bash
# The round's measurement sequence on the GPU box, from the repo root:   bash tools/run_round.sh <tag>   (e.g. r05)
# Everything lands in gpurun_out/<tag>_*; the files quoted in DESIGN.md / README.md are then copied to profiles/.
#   bench (default = one HIP graph per step, incl. the CPU baseline leg), eager, refine off, Swin;
#   rocprofv3 kernel stats of the eager step summed per kernel family;  PMC of the MSDeformAttn kernels on the op bench
#   and on the training step's own launches;  op-level benches of the gather / GEMMs / convolutions / attention;
#   the one-rank RCCL runs of the data-parallel step (eager hooks and captured exchange).
set -x
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
T=${1:-r05}
O=gpurun_out
mkdir -p $O
one() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['unit'], d['ms_per_step'], 'ms', d['config']['step_mode'], (d.get('roofline') or {}).get('frac'))" "$1"; }
python bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err; one $O/${T}_bench_default.json
python bench.py --graph --no-cpu-baseline > $O/${T}_bench_graph.json 2> /dev/null; one $O/${T}_bench_graph.json
python bench.py --no-graph --no-cpu-baseline > $O/${T}_bench_eager.json 2> /dev/null; one $O/${T}_bench_eager.json
python bench.py --graph --no-refine --no-cpu-baseline > $O/${T}_bench_no_refine.json 2> /dev/null; one $O/${T}_bench_no_refine.json
bash tools/profile_step.sh ${T} > /dev/null 2>&1; head -12 $O/${T}_bench_eager_by_family.txt
bash tools/collect_msda_pmc.sh $O/${T}_msda_pmc > /dev/null 2>&1; cp $O/${T}_msda_pmc/summary.json $O/${T}_msda_pmc.json
CMD="env TRANSOAR_BENCH_SKIP_OTHER=1 python bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline" PASSES="1 2 3" bash tools/collect_msda_pmc.sh $O/${T}_msda_pmc_step > /dev/null 2>&1; cp $O/${T}_msda_pmc_step/summary.json $O/${T}_msda_pmc_step.json
python tools/bench_msda.py --iters 20 --dtypes bf16 > $O/${T}_msda_op_bench.jsonl 2>/dev/null; cut -c1-200 $O/${T}_msda_op_bench.jsonl
python tools/bench_msda.py --iters 20 --dtypes bf16 --proj > $O/${T}_msda_op_bench_proj.jsonl 2>/dev/null
python tools/check_pcm.py --dists model,init,uniform,wide > $O/${T}_msda_fwd_kernels.jsonl 2>/dev/null; tail -3 $O/${T}_msda_fwd_kernels.jsonl
python tools/bench_gemm.py > $O/${T}_gemm_bench.jsonl 2>/dev/null
python tools/bench_convgemm.py --no-miopen > $O/${T}_conv_layers_own.jsonl 2>/dev/null
python tools/bench_convgemm.py > $O/${T}_conv_layers.jsonl 2>/dev/null; cut -c1-220 $O/${T}_conv_layers.jsonl | head -4
TRANSOAR_FORCE_DP=1 python bench.py --no-cpu-baseline --no-graph --steps 20 --warmup 5 > $O/${T}_bench_one_rank_rccl.json 2>/dev/null; one $O/${T}_bench_one_rank_rccl.json
TRANSOAR_FORCE_DP=1 python bench.py --no-cpu-baseline --graph --steps 20 --warmup 5 > $O/${T}_bench_one_rank_rccl_graph.json 2>/dev/null; one $O/${T}_bench_one_rank_rccl_graph.json
python bench.py --graph --swin --no-refine --no-cpu-baseline > $O/${T}_bench_swin.json 2> /dev/null; one $O/${T}_bench_swin.json
python bench.py --graph --swin --no-cpu-baseline > $O/${T}_bench_swin_refine.json 2> /dev/null; one $O/${T}_bench_swin_refine.json
bash tools/profile_step.sh ${T}_swin --swin --no-refine > /dev/null 2>&1; cp $O/${T}_swin_bench_eager_by_family.txt $O/${T}_bench_swin_by_family.txt; head -12 $O/${T}_bench_swin_by_family.txt
python tools/bench_roi_attn.py > $O/${T}_roi_attn_bench.jsonl 2>/dev/null; tail -3 $O/${T}_roi_attn_bench.jsonl | cut -c1-220
(python tools/bench_win_attn.py; python tools/bench_win_attn.py --shifted) > $O/${T}_win_attn_bench.jsonl 2>/dev/null; head -2 $O/${T}_win_attn_bench.jsonl | cut -c1-220
python tools/bench_gelu_mlp.py > $O/${T}_gelu_mlp_bench.jsonl 2>/dev/null; head -1 $O/${T}_gelu_mlp_bench.jsonl | cut -c1-300
# (the raw counter files of these two passes are tens of MB: they stay in /tmp, only the per-kernel summaries come back)
(cd /tmp; bash $OLDPWD/tools/pmc_any.sh /tmp/${T}_pmc_all "" -- env TRANSOAR_BENCH_SKIP_OTHER=1 python $OLDPWD/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/${T}_pmc_all.txt 2>&1); python tools/pmc_rank.py $O/${T}_pmc_all.txt > $O/${T}_pmc_rank.txt; head -6 $O/${T}_pmc_rank.txt
(cd /tmp; bash $OLDPWD/tools/pmc_any.sh /tmp/${T}_pmc_swin_all "" -- env TRANSOAR_BENCH_SKIP_OTHER=1 python $OLDPWD/bench.py --swin --no-refine --no-graph --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/${T}_pmc_swin_all.txt 2>&1); python tools/pmc_rank.py $O/${T}_pmc_swin_all.txt > $O/${T}_pmc_rank_swin.txt
[ -n "${SKIP_CPU_STEP:-}" ] || { python bench.py --cpu-baseline-only --cpu-baseline-step > $O/${T}_cpu_step.json 2>/dev/null; tail -1 $O/${T}_cpu_step.json | cut -c1-300; }
