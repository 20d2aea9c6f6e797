"""Per-queue kernel time of a rocprofv3 kernel trace (csv), per bench step: which kernels make up the main stream's critical path and
which the side stream's.   python scripts/queue_breakdown.py <kernel_trace.csv> <steps in the kept part> [skip_fraction=0.3]"""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '?')),
                     r.get('Grid_Size', r.get('Grid_Size_X', ''))))
rows.sort()
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
T0, T1 = rows[0][0], max(r[1] for r in rows)
cut = T0 + (T1 - T0) * skip
rows = [r for r in rows if r[0] >= cut]
span = (max(r[1] for r in rows) - rows[0][0]) / 1e3
print('kept span %.1f us (%.1f us per step at %g steps)' % (span, span / steps, steps))
byq = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for a, b, name, q, grid in rows:
    n = re.sub(r'\(.*', '', name).replace('void pivp::', '').replace('pivp::', '')
    e = byq[q][n[:60]]
    e[0] += 1; e[1] += (b - a) / 1e3
for q in sorted(byq):
    tot = sum(v[1] for v in byq[q].values())
    print('queue %s: %.1f us of kernels per step, %d launches per step' % (q, tot / steps, sum(v[0] for v in byq[q].values()) / steps))
    for n, (c, t) in sorted(byq[q].items(), key=lambda kv: -kv[1][1])[:18]:
        print('   %-60s %6.1f launches  %8.1f us per step  %6.1f us each' % (n, c / steps, t / steps, t / c))
