set -u
mkdir -p gpurun_out
run() {
  tag=$1; shift
  env "$@" python bench.py --cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --kernel-timing-steps 0 --steps 40 > gpurun_out/r04q_$tag.json 2> gpurun_out/r04q_$tag.log
  python -c "
import json; d=json.load(open('gpurun_out/r04q_$tag.json')); print('$tag', d['ms_per_step'], d.get('ms_per_step_median'))" || tail -5 gpurun_out/r04q_$tag.log
}
run all X=1
run noreal MSMC_REAL_FORK=0
run all2 X=1
run noreal2 MSMC_REAL_FORK=0
BURN=20 TRAINERS=4 timeout 600 python tools/many_trainers_probe.py 2>&1 | grep -v "amdgpu.ids\|^  File\|^Extension" | tail -2 | cut -c1-200
