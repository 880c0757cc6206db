set -u
mkdir -p gpurun_out
for n in 4 8 12 16; do
MSMC_WGRAD_STREAMS=$n python bench.py --cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --kernel-timing-steps 0 --steps 40 > gpurun_out/r04s_streams$n.json 2> gpurun_out/r04s_streams$n.log
python -c "
import json; d=json.load(open('gpurun_out/r04s_streams$n.json')); print('wgrad streams', $n, d['ms_per_step'], d.get('ms_per_step_median'))"
done
MSMC_WGRAD_BATCH=4 python bench.py --cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --kernel-timing-steps 0 --steps 40 > gpurun_out/r04s_b4.json 2> gpurun_out/r04s_b4.log
python -c "
import json; d=json.load(open('gpurun_out/r04s_b4.json')); print('batch 4', d['ms_per_step'], d.get('ms_per_step_median'))"
