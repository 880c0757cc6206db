# Runs ON the GPU box (via gpurun): rocprofv3 --kernel-trace --stats of one command, top kernels as text.
#   tools/kernel_stats.sh <out.txt> <command ...>
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_ks
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- "$@" > /tmp/prof_ks.log 2>&1 < /dev/null
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/prof_ks/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
out = open(sys.argv[1], 'w')
out.write('total kernel time %.3f ms in %d launches\n' % (tot / 1e6, sum(int(r['Calls']) for r in rows)))
for r in rows[:60]:
    out.write('%9.3f ms %6s calls %9.1f us  %5.1f%%  %s\n' % (float(r['TotalDurationNs']) / 1e6, r['Calls'], float(r['AverageNs']) / 1e3,
                                                          float(r['Percentage']), r['Name'][:120]))
PY
