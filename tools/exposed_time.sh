# Runs ON the GPU box (via gpurun): which kernels is the step WAITING for?  rocprofv3 --kernel-trace of one command, then over the
# window of the last <steps> optimizer launches: wall time, time with no kernel running (dependency / launch gaps), and per kernel
# symbol the time it ran ALONE (nothing else on the chip: the step cannot be shorter than the sum of those) next to its total.
# The window is the FASTEST run of <steps> consecutive steps in the trace (the timed graph replays, not the instrumented steps), a
# step ending with the <per step>-th launch of the marker kernel.
#   tools/exposed_time.sh <out.txt> <marker kernel substring> <marker launches per step> <steps> <command ...>
OUT=$1; MARK=$2; PER=$3; STEPS=$4; shift 4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_et
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_et -- "$@" > /tmp/prof_et.log 2>&1 < /dev/null
python - "$OUT" "$MARK" "$PER" "$STEPS" <<'PY'
import csv, glob, sys, collections
out, mark, per, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
f = glob.glob('/tmp/prof_et/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
marks = [i for i, r in enumerate(rows) if mark in r[2]]
best = None
for a in range(0, len(marks) - per * steps):
    w = rows[marks[a + per * steps]][1] - rows[marks[a]][1]
    if best is None or w < best[0]: best = (w, a)
a = best[1]
lo, hi = rows[marks[a]][1], rows[marks[a + per * steps]][1]
win = [r for r in rows if r[0] >= lo and r[1] <= hi]
ev = []
for k, (s, e, n) in enumerate(win):
    ev.append((s, 1, k)); ev.append((e, 0, k))
ev.sort()
running = set(); last = lo; idle = 0; alone = collections.Counter(); total = collections.Counter(); conc = collections.Counter()
for t, kind, k in ev:
    dt = t - last
    if dt > 0:
        if not running: idle += dt
        elif len(running) == 1: alone[win[next(iter(running))][2]] += dt
        conc[min(len(running), 4)] += dt
    last = t
    if kind: running.add(k)
    else: running.discard(k)
for s, e, n in win: total[n] += e - s
wall = hi - lo
o = open(out, 'w')
o.write('window: %d steps, %.3f ms per step; %d launches per step\n' % (steps, wall / steps / 1e6, len(win) / steps))
o.write('per step: no kernel running %.3f ms, one %.3f ms, two %.3f ms, three %.3f ms, four or more %.3f ms\n' % tuple(conc[i] / steps / 1e6 for i in range(5)))
o.write('kernel time per step (sum over streams) %.3f ms\n' % (sum(total.values()) / steps / 1e6))
o.write('%10s %10s  symbol (ms per step: running alone | total)\n' % ('alone', 'total'))
for n, v in sorted(total.items(), key=lambda kv: -alone[kv[0]])[:45]:
    o.write('%10.3f %10.3f  %s\n' % (alone[n] / steps / 1e6, v / steps / 1e6, n[:110]))
PY
