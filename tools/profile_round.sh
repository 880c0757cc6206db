#!/bin/bash
# Runs ON the GPU box (via gpurun): rocprofv3 kernel stats of the default bench command, then the two PMC passes
# (counters in their own runs, kernel-trace only).  Summaries land in gpurun_out/ (copied to profiles/ by hand).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
TAG=${1:-r01}
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $REPO/bench.py < /dev/null > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_bench_under_rocprof.log
cp $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
SHORT="--steps 2 --warmup 2 --no-microbench --cpu-steps 0 --kernel-timing-steps 0 --fp32-steps 0"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- python $REPO/bench.py $SHORT > /dev/null 2> $OUT/${TAG}_pmc_fetch.log
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- python $REPO/bench.py $SHORT > /dev/null 2> $OUT/${TAG}_pmc_write.log
timeout 900 rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d /tmp/prof_mfma -- python $REPO/bench.py $SHORT > /dev/null 2> $OUT/${TAG}_pmc_mfma.log
# the VQ argmin micro-benchmark at N = 2^20 frames (the half of the BASELINE metric the step does not exercise at size);
# SKIP_VQ=1 keeps the committed passes (the VQ kernels did not change)
if [ -z "${SKIP_VQ:-}" ]; then
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_vq_fetch -- python $REPO/bench.py --microbench-only > /dev/null 2> $OUT/${TAG}_pmc_vq_fetch.log
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_vq_write -- python $REPO/bench.py --microbench-only > /dev/null 2> $OUT/${TAG}_pmc_vq_write.log
python $REPO/tools/pmc_summary.py $OUT/${TAG}_pmc_vq_traffic.json FETCH=/tmp/prof_vq_fetch WRITE=/tmp/prof_vq_write > $OUT/${TAG}_pmc_vq_summary.txt 2>&1
fi
python $REPO/tools/pmc_summary.py $OUT/${TAG}_pmc_traffic.json FETCH=/tmp/prof_fetch WRITE=/tmp/prof_write MFMA=/tmp/prof_mfma > $OUT/${TAG}_pmc_summary.txt 2>&1
cd $REPO
cp $OUT/${TAG}_pmc_traffic.json $REPO/profiles/pmc_traffic.json      # the line below reports roofline.traffic from this run's passes
python bench.py --calls-out $OUT/${TAG}_layer_table.json --kernels-out $OUT/${TAG}_bench_kernels.json --steps 50 --warmup 10 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.log
tail -c 1500 $OUT/${TAG}_bench.json
head -12 $OUT/${TAG}_pmc_summary.txt
