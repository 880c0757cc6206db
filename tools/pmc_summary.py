#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel symbol.

usage: pmc_summary.py out.json FETCH=<dir-or-csv> WRITE=<dir-or-csv> [MFMA=<dir-or-csv>]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of the TCC EA request counters; following
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE is doubled on gfx950 (128-byte requests tallied at
64 B); WRITE_SIZE is taken as reported (uncalibrated).  Output: bytes per launch per kernel symbol."""
import csv, glob, json, os, re, sys, collections


def short(name):
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*$', '', name).strip()


def load(path, counter):
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True)
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') != counter:
                continue
            k = short(row['Kernel_Name'])
            acc[k][0] += 1
            acc[k][1] += float(row['Counter_Value'])
    return acc


def main():
    out = sys.argv[1]
    args = dict(a.split('=', 1) for a in sys.argv[2:])
    fetch = load(args['FETCH'], 'FETCH_SIZE') if 'FETCH' in args else {}
    write = load(args['WRITE'], 'WRITE_SIZE') if 'WRITE' in args else {}
    mfma = load(args['MFMA'], 'MfmaUtil') if 'MFMA' in args else {}
    kernels = {}
    for k in sorted(set(fetch) | set(write) | set(mfma)):
        nf, sf = fetch.get(k, (0, 0.0))
        nw, sw = write.get(k, (0, 0.0))
        fetch_b = 2.0 * 1024.0 * sf / nf if nf else None      # KiB -> bytes, x2 gfx950 correction
        write_b = 1024.0 * sw / nw if nw else None
        kernels[k] = dict(launches_fetch_pass=nf, launches_write_pass=nw, fetch_bytes_per_launch=fetch_b,
                          write_bytes_per_launch=write_b, raw_fetch_size_sum=sf, raw_write_size_sum=sw,
                          hbm_bytes_per_launch=(fetch_b or 0.0) + (write_b or 0.0),
                          mfma_util_percent=(mfma[k][1] / mfma[k][0]) if k in mfma and mfma[k][0] else None)
    json.dump(dict(note='rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950 '
                        'correction, MI355X_MICROARCH.md), unit KiB -> bytes; WRITE_SIZE uncalibrated',
                   kernels=kernels), open(out, 'w'), indent=1, sort_keys=True)
    top = sorted(kernels.items(), key=lambda kv: -(kv[1]['launches_fetch_pass'] or 0))[:25]
    for k, v in top:
        print('%-70s n=%5d fetch %10.0f B  write %10.0f B  MfmaUtil %s' % (k[:70], v['launches_fetch_pass'], v['fetch_bytes_per_launch'] or 0,
              v['write_bytes_per_launch'] or 0, '%.2f %%' % v['mfma_util_percent'] if v['mfma_util_percent'] is not None else '-'))


if __name__ == '__main__':
    main()
