set -u
mkdir -p gpurun_out
python tools/torch_ops_in_step.py > gpurun_out/r04_stock_operators_in_step.txt 2>&1
python bench.py --calls-out gpurun_out/r04_layer_table.json --kernels-out gpurun_out/r04_bench_kernels.json --steps 50 --warmup 10 > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.log
python -c "
import json
d=json.load(open('gpurun_out/r04_bench.json')); print(d['value'], d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['frac'], d['roofline_step'])"
grep "in total per step" gpurun_out/r04_stock_operators_in_step.txt
