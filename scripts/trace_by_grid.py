"""Per-dispatch durations grouped by (kernel, grid) from a rocprofv3 kernel trace:
python scripts/trace_by_grid.py <kernel_trace.csv> <substring of the kernel name> [...]"""
import csv, sys, collections
acc = collections.defaultdict(list)
pats = sys.argv[2:]
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if any(p in n for p in pats):
        grid = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
        acc[(n[:48], grid)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print('%-50s blocks %-16s n %4d  avg %7.1f us  total %7.2f ms' % (k[0], k[1], len(v), sum(v) / len(v), sum(v) / 1e3))
