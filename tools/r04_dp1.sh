# world-size-1 data-parallel step under torchrun against the plain step: where do the extra ~2 ms go?
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ARGS="--gpus 1 --steps 20 --warmup 5 --cpu-steps 0 --fp32-steps 0 --no-microbench --warmup-phase-steps 0 --kernel-timing-steps 0"
cd /tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dp1 -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 $R/bench.py $ARGS > $R/gpurun_out/r04_dp1_bench.json 2> $R/gpurun_out/r04_dp1_bench.log
f=$(find /tmp/prof_dp1 -name '*kernel_stats.csv' | xargs ls -S | head -1)
cp $f $R/gpurun_out/r04_dp1_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dp0 -- python $R/bench.py $ARGS > $R/gpurun_out/r04_dp0_bench.json 2> $R/gpurun_out/r04_dp0_bench.log
f=$(find /tmp/prof_dp0 -name '*kernel_stats.csv' | xargs ls -S | head -1)
cp $f $R/gpurun_out/r04_dp0_kernel_stats.csv
tail -c 300 $R/gpurun_out/r04_dp1_bench.log; tail -c 300 $R/gpurun_out/r04_dp0_bench.log
