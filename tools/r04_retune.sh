set -u
mkdir -p gpurun_out
( time RETUNE=all python tools/tune_bench_shapes.py ) > gpurun_out/r04_retune.log 2>&1
tail -5 gpurun_out/r04_retune.log
cp gpurun_out/tuned_gfx950.json msmc-tts_amd/msmctts_amd/hip/tuned_gfx950.json
python bench.py --kernels-out gpurun_out/r04n_bench_kernels.json --cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --steps 30 > gpurun_out/r04n_bench.json 2> gpurun_out/r04n_bench.log
python -c "
import json; d=json.load(open('gpurun_out/r04n_bench.json')); print(d['ms_per_step'], d['ms_per_step_median'], d['roofline_step']['frac'])"
