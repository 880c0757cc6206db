set -u
mkdir -p gpurun_out
run() {
  tag=$1; shift
  env "$@" python bench.py --cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --kernel-timing-steps 0 --steps 30 > gpurun_out/r04k_$tag.json 2> gpurun_out/r04k_$tag.log
  python -c "
import json; d=json.load(open('gpurun_out/r04k_$tag.json')); print('$tag', d['ms_per_step'], d.get('ms_per_step_median'))" || tail -5 gpurun_out/r04k_$tag.log
}
run fork MSMC_D_FORK=1
run fork_w2 MSMC_D_FORK=1 MSMC_WGRAD_STREAMS=2
run fork_w4 MSMC_D_FORK=1 MSMC_WGRAD_STREAMS=4
MSMC_D_FORK=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
