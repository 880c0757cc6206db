set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
SHORT="--steps 2 --warmup 2 --no-microbench --cpu-steps 0 --kernel-timing-steps 0 --fp32-steps 0 --warmup-phase-steps 0"
rm -f $R/gpurun_out/r04_step_sq_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES"; do
  rm -rf /tmp/pmc_step
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_step -- python $R/bench.py $SHORT > /dev/null 2> /tmp/pmc_step.log
  python $R/tools/pmc_sq.py /tmp/pmc_step >> $R/gpurun_out/r04_step_sq_counters.txt 2>&1
done
wc -l $R/gpurun_out/r04_step_sq_counters.txt
