# Runs ON the GPU box (via gpurun): SQ counter passes of one command, per-kernel averages for the symbols matching <filter>.
#   tools/sq_pass.sh <out.txt> <filter> <command ...>        (counters in their own runs, kernel-trace only)
OUT=$1; FLT=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $OUT
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/prof_sq
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/prof_sq -- "$@" > /tmp/prof_sq.log 2>&1 < /dev/null || echo "pass failed: $SET" >> $OUT
  python tools/pmc_sq.py /tmp/prof_sq "$FLT" >> $OUT 2>&1
done
