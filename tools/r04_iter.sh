set -u
mkdir -p gpurun_out
TAG=${1:-r04f}
VARIANTS="50 2" python tools/bench_gather3.py thin > gpurun_out/${TAG}_thin_microbench.txt 2>&1
python -m pytest tests/test_gpu_conv.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
python bench.py --kernels-out gpurun_out/${TAG}_bench_kernels.json --cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --steps 30 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.log
cat gpurun_out/${TAG}_thin_microbench.txt
python -c "
import json,sys
d=json.load(open('gpurun_out/${TAG}_bench.json')); print(d['ms_per_step'], d['ms_per_step_median'], d['roofline_step']['frac'])
k=json.load(open('gpurun_out/${TAG}_bench_kernels.json'))['kernels']
for n,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step'])[:22]: print('%.3f %5.1f %6.1f %s' % (v['ms_per_step'], v['launches']/3, v['avg_us'], n[:60]))
"
