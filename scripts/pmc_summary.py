"""Summarise rocprofv3 --pmc passes (one counter per pass, as MI355X_MICROARCH.md prescribes) into per-kernel averages and
the HBM-traffic figure bench.py reports as roofline.traffic.

  python scripts/pmc_summary.py <fetch_dir> <write_dir> <out_dir> ["workload description"]

<fetch_dir>/<write_dir>: output directories of `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` with --output-format csv.
Writes <out_dir>/pmc_fetch_size_summary.csv, pmc_write_size_summary.csv and pmc_traffic.json.
FETCH_SIZE is doubled for the traffic figure: on gfx950 it reports half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(d, counter):
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    assert files, 'no counter_collection.csv under ' + d
    acc = defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') != counter:
                continue
            a = acc[r['Kernel_Name']]
            a[0] += 1; a[1] += float(r['Counter_Value'])
    return {k: (n, v / n) for k, (n, v) in acc.items()}


def write_summary(path, rows, col):
    with open(path, 'w') as f:
        f.write('kernel,launches,%s\n' % col)
        for k, (n, v) in sorted(rows.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
            f.write('"%s",%d,%.1f\n' % (k[:70], n, v))


def main():
    fetch_dir, write_dir, out_dir = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else ''
    fe = per_kernel(fetch_dir, 'FETCH_SIZE')
    wr = per_kernel(write_dir, 'WRITE_SIZE')
    os.makedirs(out_dir, exist_ok=True)
    write_summary(os.path.join(out_dir, 'pmc_fetch_size_summary.csv'), fe, 'avg_FETCH_SIZE_KB')
    write_summary(os.path.join(out_dir, 'pmc_write_size_summary.csv'), wr, 'avg_WRITE_SIZE_KB')
    is_lstm = lambda k: 'igemm_f32_kernel' in k and 'true' in k
    nf = sum(n for k, (n, v) in fe.items() if is_lstm(k)); f_kb = sum(n * v for k, (n, v) in fe.items() if is_lstm(k)) / max(nf, 1)
    nw = sum(n for k, (n, v) in wr.items() if is_lstm(k)); w_kb = sum(n * v for k, (n, v) in wr.items() if is_lstm(k)) / max(nw, 1)
    out = {
        'kernel': 'igemm_f32_kernel<*,*,4,true> (ConvLSTM)',
        'workload': workload,
        'launches': nf,
        'fetch_size_KB_raw': f_kb,
        'write_size_KB': w_kb,
        'hbm_bytes_per_launch': (2.0 * f_kb + w_kb) * 1024.0,
        'note': 'separate --pmc passes (FETCH_SIZE, WRITE_SIZE); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of '
                'wide coalesced reads; uncalibrated for this gather); Infinity-Cache hits are counted',
    }
    json.dump(out, open(os.path.join(out_dir, 'pmc_traffic.json'), 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
