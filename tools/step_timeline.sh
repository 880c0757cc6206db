# Runs ON the GPU box (via gpurun): the kernels of ONE replayed step in start order -- start (us from the step's first kernel),
# duration, number of other kernels running when it starts, gap since the previous kernel END on the chip, symbol.
# The step is the fastest window between two consecutive <per step>-th launches of the marker kernel in a rocprofv3 kernel trace.
#   tools/step_timeline.sh <out.txt> <marker kernel substring> <marker launches per step> <command ...>
OUT=$1; MARK=$2; PER=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_tl
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -- "$@" > /tmp/prof_tl.log 2>&1 < /dev/null
python - "$OUT" "$MARK" "$PER" <<'PY'
import csv, glob, sys
out, mark, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = glob.glob('/tmp/prof_tl/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', ''))) for r in csv.DictReader(open(f))]
rows.sort()
marks = [i for i, r in enumerate(rows) if mark in r[2]]
best = None
for a in range(0, len(marks) - per):
    w = rows[marks[a + per]][1] - rows[marks[a]][1]
    if best is None or w < best[0]: best = (w, a)
a = best[1]
lo, hi = rows[marks[a]][1], rows[marks[a + per]][1]
win = [r for r in rows if r[0] >= lo and r[1] <= hi]
o = open(out, 'w')
o.write('one step: %.3f ms, %d launches\n' % ((hi - lo) / 1e6, len(win)))
o.write('%9s %8s %4s %7s %6s  symbol\n' % ('start_us', 'dur_us', 'conc', 'gap_us', 'queue'))
ends = []
last_end = lo
for s, e, n, q in win:
    conc = sum(1 for x in ends if x > s)
    gap = (s - last_end) / 1e3 if conc == 0 else 0.0
    o.write('%9.1f %8.1f %4d %7.1f %6s  %s\n' % ((s - lo) / 1e3, (e - s) / 1e3, conc, gap, q, n[:100]))
    ends = [x for x in ends if x > s] + [e]
    last_end = max(last_end, e)
PY
