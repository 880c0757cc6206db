set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
VARIANTS="50 2 3 8" python tools/bench_gather3.py thin > gpurun_out/r04_thin_microbench.txt 2>&1
cd /tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES"; do
  rm -rf /tmp/pmc_thin
  VARIANTS="50" rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_thin -- python $R/tools/bench_gather3.py "thin mrd 8->16" > /dev/null 2> /tmp/pmc_thin.log
  python $R/tools/pmc_sq.py /tmp/pmc_thin conv_gather6 >> $R/gpurun_out/r04_thin_sq_counters.txt 2>&1
done
cat $R/gpurun_out/r04_thin_microbench.txt
cat $R/gpurun_out/r04_thin_sq_counters.txt
