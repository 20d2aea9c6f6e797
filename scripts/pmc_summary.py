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


def mfma_busy(d, out_path):
    """MFMA-pipe utilisation per kernel from a `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass:
    busy cycles summed over the 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs x 1024)."""
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    assert files, 'no counter_collection.csv under ' + d
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    rows = []
    for k, v in acc.items():
        if not v.get('SQ_VALU_MFMA_BUSY_CYCLES') or not v.get('GRBM_GUI_ACTIVE'):
            continue
        b = sum(v['SQ_VALU_MFMA_BUSY_CYCLES']) / len(v['SQ_VALU_MFMA_BUSY_CYCLES'])
        g = sum(v['GRBM_GUI_ACTIVE']) / len(v['GRBM_GUI_ACTIVE'])
        if b > 0:
            rows.append((k, len(v['GRBM_GUI_ACTIVE']), b, g, b / (g / 8.0 * 1024.0)))
    with open(out_path, 'w') as f:
        f.write('kernel,launches,avg_SQ_VALU_MFMA_BUSY_CYCLES,avg_GRBM_GUI_ACTIVE,mfma_busy_fraction\n')
        for k, n, b, g, u in sorted(rows, key=lambda r: -r[1] * r[3]):
            f.write('"%s",%d,%.4e,%.4e,%.3f\n' % (k[:70], n, b, g, u))
    return rows


def main():
    if sys.argv[1] == '--mfma':          # python scripts/pmc_summary.py --mfma <pmc_dir> <out_csv>
        for r in mfma_busy(sys.argv[2], sys.argv[3])[:6]:
            print('%-60s launches %4d  MFMA busy fraction %.3f' % (r[0][:60], r[1], r[4]))
        return
    bf16 = sys.argv[1] == '--bf16'       # python scripts/pmc_summary.py --bf16 <fetch_dir> <write_dir> <out_dir> [workload]
    if bf16:
        del sys.argv[1]
    fetch_dir, write_dir, out_dir = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else ''
    fe = per_kernel(fetch_dir, 'FETCH_SIZE')
    wr = per_kernel(write_dir, 'WRITE_SIZE')
    os.makedirs(out_dir, exist_ok=True)
    sfx = '_bf16' if bf16 else ''
    write_summary(os.path.join(out_dir, 'pmc_fetch_size_summary%s.csv' % sfx), fe, 'avg_FETCH_SIZE_KB')
    write_summary(os.path.join(out_dir, 'pmc_write_size_summary%s.csv' % sfx), wr, 'avg_WRITE_SIZE_KB')
    is_lstm = (lambda k: 'convlstm_bf16_kernel' in k and 'true' in k) if bf16 else (lambda k: 'igemm_f32_kernel' in k and 'true' in k)
    nf = sum(n for k, (n, v) in fe.items() if is_lstm(k)); f_kb = sum(n * v for k, (n, v) in fe.items() if is_lstm(k)) / max(nf, 1)
    nw = sum(n for k, (n, v) in wr.items() if is_lstm(k)); w_kb = sum(n * v for k, (n, v) in wr.items() if is_lstm(k)) / max(nw, 1)
    out = {
        'kernel': 'convlstm_bf16_kernel<*,true,1> (ConvLSTM, bf16 operands)' if bf16 else 'igemm_f32_kernel<*,*,4,true> (ConvLSTM)',
        'workload': workload,
        'launches': nf,
        'fetch_size_KB_raw': f_kb,
        'write_size_KB': w_kb,
        'hbm_bytes_per_launch': (2.0 * f_kb + w_kb) * 1024.0,
        'note': 'separate --pmc passes (FETCH_SIZE, WRITE_SIZE); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of '
                'wide coalesced reads; uncalibrated for this gather); Infinity-Cache hits are counted',
    }
    json.dump(out, open(os.path.join(out_dir, 'pmc_traffic%s.json' % sfx), 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
