set -u
mkdir -p gpurun_out
( time RETUNE=new python tools/tune_bench_shapes.py ) > gpurun_out/r04_retune2.log 2>&1
tail -4 gpurun_out/r04_retune2.log
cp gpurun_out/tuned_gfx950.json msmc-tts_amd/msmctts_amd/hip/tuned_gfx950.json
run() {
  tag=$1; shift
  env "$@" python bench.py --cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --kernel-timing-steps 0 --steps 40 > gpurun_out/r04t_$tag.json 2> gpurun_out/r04t_$tag.log
  python -c "
import json; d=json.load(open('gpurun_out/r04t_$tag.json')); print('$tag', d['ms_per_step'], d.get('ms_per_step_median'))" || tail -5 gpurun_out/r04t_$tag.log
}
run fork X=1
run nofork MSMC_FFT_FORK=0
run fork2 X=1
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
