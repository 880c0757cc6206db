set -u
mkdir -p gpurun_out
SKIP_VQ=1 bash tools/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > gpurun_out/r04_gpu_suite.txt 2>&1
SHORT="--cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --kernel-timing-steps 0 --steps 30"
for c in 1 3 5; do python bench.py --config $c $SHORT > gpurun_out/r04_bench_config$c.json 2> gpurun_out/r04_bench_config$c.log; done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bf16 -- python bench.py --fp32-steps 0 --no-microbench --cpu-steps 0 --kernel-timing-steps 0 > gpurun_out/r04_bf16_step_under_rocprof.json 2> gpurun_out/r04_bf16_step_under_rocprof.log
cp $(find /tmp/prof_bf16 -name '*kernel_stats.csv' | head -1) gpurun_out/r04_kernel_stats_bf16_step.csv
python tools/torch_ops_in_step.py > gpurun_out/r04_stock_operators_in_step.txt 2>&1
tail -3 gpurun_out/r04_gpu_suite.txt
python -c "
import json
for f in ('r04_bench','r04_bench_config1','r04_bench_config3','r04_bench_config5'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d.get('ms_per_step_median'), (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('traffic'), (d.get('roofline_step') or {}).get('frac'))
"
