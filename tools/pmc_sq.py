#!/usr/bin/env python
"""Per-kernel averages of the SQ counters of one rocprofv3 --pmc pass (counter_collection CSVs under the given directory)."""
import collections, csv, glob, os, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r'\(.*$', '', re.sub(r'^void ', '', row['Kernel_Name'])).strip()
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        a = acc[k + ' grid=' + row.get('Grid_Size', '?')][row['Counter_Name']]
        a[0] += 1
        a[1] += float(row['Counter_Value'])
for k, cs in sorted(acc.items()):
    print(k)
    base = cs.get('SQ_WAVE_CYCLES', [1, 0.0])
    for c, (n, v) in sorted(cs.items()):
        print('   %-28s n=%3d avg %16.0f  %s' % (c, n, v / n, ('%.1f %% of SQ_WAVE_CYCLES' % (100.0 * v / base[1])) if base[1] and c.startswith('SQ_') else ''))
