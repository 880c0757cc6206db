set -u
mkdir -p gpurun_out
run() {
  tag=$1; shift
  env "$@" python bench.py --cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --kernel-timing-steps 0 --steps 30 > gpurun_out/r04k_$tag.json 2> gpurun_out/r04k_$tag.log
  python -c "
import json; d=json.load(open('gpurun_out/r04k_$tag.json')); print('$tag', d['ms_per_step'], d.get('ms_per_step_median'))" || tail -5 gpurun_out/r04k_$tag.log
}
run fork_w8 MSMC_D_FORK=1 MSMC_WGRAD_STREAMS=8
run fork_w3 MSMC_D_FORK=1 MSMC_WGRAD_STREAMS=3
run fork_w4_b1 MSMC_D_FORK=1 MSMC_WGRAD_STREAMS=4 MSMC_WGRAD_BATCH=1
run fork_w4_b4 MSMC_D_FORK=1 MSMC_WGRAD_STREAMS=4 MSMC_WGRAD_BATCH=4
run fork_w4_b16 MSMC_D_FORK=1 MSMC_WGRAD_STREAMS=4 MSMC_WGRAD_BATCH=16
run fork_w4_pc MSMC_D_FORK=1 MSMC_WGRAD_STREAMS=4 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
