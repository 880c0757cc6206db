set -u
mkdir -p gpurun_out
for n in 1 2 4; do
MSMC_WGRAD_STREAMS=$n python bench.py --cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --kernel-timing-steps 0 --steps 30 > gpurun_out/r04k_streams$n.json 2> gpurun_out/r04k_streams$n.log
python -c "
import json; d=json.load(open('gpurun_out/r04k_streams$n.json')); print('wgrad streams', $n, d['ms_per_step'], d.get('ms_per_step_median'))"
done
