# gpurun command line of round 5's kernel iterations (run from the repository root on the GPU box):
#   tools/r05_iter.sh <tag> [micro] [conv] [tune] [bench]
# micro: forward-convolution micro-benchmark, fifth against seventh generation + ablations of the seventh
# conv:  GPU convolution tests (forced-candidate sweeps included)
# tune:  RETUNE=<mode> tools/tune_bench_shapes.py (the table stays in gpurun_out/ and is used by the bench of the same call)
# bench: 30-step bench with the per-kernel table
set -u
mkdir -p gpurun_out
TAG=${1:-r05}
shift
for what in "$@"; do
case $what in
micro)
  VARIANTS="${VARIANTS:-40 41 42 44 45 46 56 59 60 61 63}" env -u ABLATE python tools/bench_gather3.py ${FILTER:-} 2>&1 | grep -v "^thin" > gpurun_out/${TAG}_gather7_microbench.txt
  ABLATE="${ABLATE:-56 59 63}" ABLS="0 2 4 8 32 46" python tools/bench_gather3.py ffn >> gpurun_out/${TAG}_gather7_microbench.txt 2>&1
  cat gpurun_out/${TAG}_gather7_microbench.txt ;;
conv)
  timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -5 ;;
parity)
  timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -5 ;;
tune)
  RETUNE=${RETUNE:-gather7} python tools/tune_bench_shapes.py 2>&1 | tail -3
  export MSMC_TUNE_CACHE=$PWD/gpurun_out/tuned_gfx950.json ;;
bench)
  python bench.py --kernels-out gpurun_out/${TAG}_bench_kernels.json --calls-out gpurun_out/${TAG}_layer_table.json --cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --steps 30 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.log || tail -20 gpurun_out/${TAG}_bench.log
  python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench.json')); print('ms/step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'roofline_step', d['roofline_step']['frac'])
k=json.load(open('gpurun_out/${TAG}_bench_kernels.json'))['kernels']
for n,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step'])[:26]: print('%.3f %5.1f %6.1f %s' % (v['ms_per_step'], v['launches']/3, v['avg_us'], n[:70]))
PY
  ;;
bench4)
  python bench.py --config 4 --kernels-out gpurun_out/${TAG}_bench_config4_kernels.json --steps 10 --warmup 3 --cpu-steps ${CPU_STEPS:-2} --cpu-warmup 1 > gpurun_out/${TAG}_bench_config4.json 2> gpurun_out/${TAG}_bench_config4.log || tail -n 25 gpurun_out/${TAG}_bench_config4.log
  cut -c1-1500 gpurun_out/${TAG}_bench_config4.json ;;
esac
done
