"""One timestep of the rollout as a timeline: kernel, duration, idle gap in front of it (from a rocprofv3 kernel trace csv).
python scripts/timestep_timeline.py <kernel_trace.csv> [which]   -- `which`: index of the conv_enc0_rows launch that starts the printed timestep"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(.*', '', r['Kernel_Name']).replace('void pivp::', '').replace('pivp::', '')))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith('conv_enc0_rows')]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
i0, i1 = starts[which], starts[which + 1]
t0 = rows[i0][0]
busy = gaps = 0.0
for i in range(i0, i1):
    a, b, n = rows[i]
    gap = (a - rows[i - 1][1]) / 1e3
    busy += (b - a) / 1e3; gaps += max(gap, 0.0)
    print('%8.1f us  %-46s %7.1f us   gap before %5.1f us' % ((a - t0) / 1e3, n[:46], (b - a) / 1e3, gap))
print('timestep: %.1f us, kernels %.1f us, gaps %.1f us over %d launches' % ((rows[i1][0] - t0) / 1e3, busy, gaps, i1 - i0))
# all timesteps: mean gap per launch
tot_gap = 0.0; n = 0
for i in range(starts[2], starts[-1]):
    tot_gap += max(rows[i][0] - rows[i - 1][1], 0) / 1e3; n += 1
print('mean gap over %d launches: %.2f us' % (n, tot_gap / max(n, 1)))
