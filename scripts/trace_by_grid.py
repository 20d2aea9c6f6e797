"""Kernel trace grouped by (kernel, grid): tells the layers of one templated kernel apart.
python scripts/trace_by_grid.py <kernel_trace.csv> [steps]   -> name, grid, calls, avg us, total ms (per step when `steps` is given)"""
import csv
import re
import sys
from collections import defaultdict


def main():
    rows = defaultdict(lambda: [0, 0.0])
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void pivp::', '').replace('pivp::', '')
            grid = '%sx%sx%s' % (r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
            k = (name, grid, r.get('LDS_Block_Size', ''))
            rows[k][0] += 1
            rows[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    tot = sum(v[1] for v in rows.values())
    print('total %.3f ms%s' % (tot / 1e3 / steps, ' per step' if steps != 1 else ''))
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1])[:70]:
        print('%-52s %-16s lds %-7s calls %6.1f  avg %8.1f us  total %8.3f ms' % (k[0][:52], k[1], k[2], v[0] / steps, v[1] / v[0], v[1] / 1e3 / steps))


if __name__ == '__main__':
    main()
