set -u
mkdir -p gpurun_out
run() {
  tag=$1; shift
  env "$@" python bench.py --cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --kernel-timing-steps 0 --steps 40 > gpurun_out/r04w_$tag.json 2> gpurun_out/r04w_$tag.log
  python -c "
import json; d=json.load(open('gpurun_out/r04w_$tag.json')); print('$tag', d['ms_per_step'], d.get('ms_per_step_median'))" || tail -5 gpurun_out/r04w_$tag.log
}
run side X=1
run main MSMC_FINISH_SIDE=0
run side2 X=1
run main2 MSMC_FINISH_SIDE=0
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
