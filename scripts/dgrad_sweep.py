"""Per-dispatch durations of the data-gradient launches of one train-step profile, grouped by (tile, grid):
python scripts/dgrad_sweep.py <kernel_trace.csv>"""
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'igemm_f32_kernel' in n and 'false' in n:
        tile = n[n.index('<'):n.index('>') + 1]
        grid = (r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', ''))
        acc[(tile, grid)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print('%-22s grid %-22s n %4d  avg %7.1f us  total %7.2f ms' % (k[0], k[1], len(v), sum(v) / len(v), sum(v) / 1e3))
