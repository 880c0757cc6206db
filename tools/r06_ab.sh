# gpurun command line of round 6's step-level A/B runs (run from the repository root on the GPU box):
#   tools/r06_ab.sh <tag> "<ENV=.. ENV=..>" ["<ENV ...>" ...]     one quick bench (30 graph-replayed steps) per environment, interleaved twice
set -u
mkdir -p gpurun_out
TAG=$1; shift
for rep in 1 2; do
for envs in "$@"; do
  line=$(env $envs python bench.py --fp32-steps 0 --no-microbench --cpu-steps 0 --kernel-timing-steps 0 --warmup-phase-steps 0 --steps 30 --warmup 8 2>gpurun_out/${TAG}_ab.log | tail -1)
  echo "$envs => $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'], 'median', d.get('ms_per_step_median'))" "$line" 2>/dev/null || (echo FAILED; tail -5 gpurun_out/${TAG}_ab.log))" | tee -a gpurun_out/${TAG}_ab.txt
done
done
