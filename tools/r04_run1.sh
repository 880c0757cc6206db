set -u
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r04a_gpu_pytest.txt 2>&1
python bench.py --calls-out gpurun_out/r04a_layer_table.json --kernels-out gpurun_out/r04a_bench_kernels.json > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.log
SHORT="--cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --kernel-timing-steps 0 --steps 30"
MSMC_FFT_PROLOGUE=1 python bench.py $SHORT > gpurun_out/r04a_bench_prologue.json 2> gpurun_out/r04a_bench_prologue.log
MSMC_SKIP_CLEAN_PREPARE=0 python bench.py $SHORT > gpurun_out/r04a_bench_noskip.json 2> gpurun_out/r04a_bench_noskip.log
python bench.py $SHORT > gpurun_out/r04a_bench_short.json 2> gpurun_out/r04a_bench_short.log
tail -5 gpurun_out/r04a_gpu_pytest.txt
for f in r04a_bench r04a_bench_prologue r04a_bench_noskip r04a_bench_short; do python -c "
import json,sys
d=json.load(open('gpurun_out/$f.json'))
print('$f', d['ms_per_step'], d.get('ms_per_step_median'), d.get('roofline_step'))
"; done
