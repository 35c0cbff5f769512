mkdir -p gpurun_out; cd /root/repo
for pb in 0 2; do
TRANSOAR_MSDA3D_Q16_PROBE=$pb bash tools/pmc_any.sh gpurun_out/pmc_q16_p$pb q16 -- python tools/check_q16.py --time-only --dists model --iters 5 > gpurun_out/pmc_q16_p$pb.txt 2>&1
cat gpurun_out/pmc_q16_p$pb.txt
done
TRANSOAR_MSDA_FLAGS=128 bash tools/pmc_any.sh gpurun_out/pmc_pcm fwd_pcm -- python tools/probe_fwd.py --dist model --iters 5 > gpurun_out/pmc_pcm.txt 2>&1; cat gpurun_out/pmc_pcm.txt
rm -rf gpurun_out/pmc_q16_p0 gpurun_out/pmc_q16_p2 gpurun_out/pmc_pcm
