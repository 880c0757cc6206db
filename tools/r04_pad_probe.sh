# What does one more instruction per stage cost gather5?  Three builds of the library: as shipped, +40 scalar instructions per
# stage and wave (CV5_SALU_PAD=40), +12 vector instructions (CV5_VALU_PAD=12); see the #if block in csrc/gather5.inc.
set -u
mkdir -p gpurun_out
OUT=gpurun_out/r04_pad_probe.txt
: > $OUT
for lib in "" msmc-tts_amd/lib/libmsmc_hip_pad_SALU.so msmc-tts_amd/lib/libmsmc_hip_pad_VALU.so; do
  echo "== ${lib:-shipped}" >> $OUT
  for f in "ffn w2 T400" "ffn w1 T400" "rb C128 L1200 k7" "mpd p2 512->512" "rb C256 L240 k3"; do
    MSMC_PROBE_LIB=$lib VARIANTS="40 41 44 45" python tools/bench_gather3.py "$f" 2>&1 | grep -v "amdgpu.ids\|^layer" >> $OUT
  done
done
cat $OUT
