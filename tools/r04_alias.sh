set -u
mkdir -p gpurun_out
echo "== own streams, fork on, pool burned"; MSMC_D_FORK=1 BURN=20 python tools/many_trainers_probe.py 2>&1 | grep -v "amdgpu.ids\|^  File\|^Extension" | tail -6
echo "== whole fullsize file, fork on"; MSMC_D_FORK=1 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -4
MSMC_D_FORK=1 python bench.py --cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --kernel-timing-steps 0 --steps 30 > gpurun_out/r04o_fork.json 2> gpurun_out/r04o_fork.log
python -c "
import json; d=json.load(open('gpurun_out/r04o_fork.json')); print('fork', d['ms_per_step'], d.get('ms_per_step_median'))"
python bench.py --cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --kernel-timing-steps 0 --steps 30 > gpurun_out/r04o_nofork.json 2> gpurun_out/r04o_nofork.log
python -c "
import json; d=json.load(open('gpurun_out/r04o_nofork.json')); print('nofork', d['ms_per_step'], d.get('ms_per_step_median'))"
