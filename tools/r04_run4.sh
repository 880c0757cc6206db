set -u
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r04d_gpu_pytest.txt 2>&1
python bench.py --calls-out gpurun_out/r04d_layer_table.json --kernels-out gpurun_out/r04d_bench_kernels.json > gpurun_out/r04d_bench.json 2> gpurun_out/r04d_bench.log
SHORT="--cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --kernel-timing-steps 0 --steps 30"
python bench.py $SHORT > gpurun_out/r04d_bench_short.json 2> gpurun_out/r04d_bench_short.log
python tools/torch_ops_in_step.py > gpurun_out/r04d_stock_operators_in_step.txt 2>&1
tail -4 gpurun_out/r04d_gpu_pytest.txt
for f in r04d_bench r04d_bench_short; do python -c "
import json,sys
d=json.load(open('gpurun_out/$f.json'))
print('$f', d['ms_per_step'], d.get('ms_per_step_median'), (d.get('roofline_step') or {}).get('frac'), (d.get('roofline_step') or {}).get('launches_per_step'), d['losses'].get('stft_loss'))
"; done
head -3 gpurun_out/r04d_stock_operators_in_step.txt
