set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_wgrad_sq_counters.txt
: > $OUT
for f in "mrd h240" "mpd p2" "mpd p11" "rb C256" "rb C128 L1200 k11" "ffn w1 T400"; do
python tools/bench_wgrad.py "$f" 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04_wgrad_microbench.txt
cd /tmp
for shape in "mrd h240 256->512 s2:7/0" "mpd p2 512->512 s1:4/0"; do
name=${shape%%:*}; cand=${shape##*:}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  rm -rf /tmp/pmc_w
  CANDS="$cand" rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_w -- python $R/tools/bench_wgrad.py "$name" > /dev/null 2> /tmp/pmc_w.log
  echo "== $name cand $cand" >> $OUT
  python $R/tools/pmc_sq.py /tmp/pmc_w conv_wgrad >> $OUT 2>&1
done
done
cat $R/gpurun_out/r04_wgrad_microbench.txt
cat $OUT
