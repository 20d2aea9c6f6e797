"""Kernel launches per bench step from a rocprofv3 kernel trace (csv): the launches between two consecutive launches of a kernel that
runs exactly once per step (adam_kernel in a train step, loss_finalize_kernel in a rollout), per kernel name.
    python scripts/launch_count.py <kernel_trace.csv> [marker substring]"""
import csv
import re
import sys
from collections import Counter

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), r['Kernel_Name']))
rows.sort()
names = [n for _, n in rows]
marker = sys.argv[2] if len(sys.argv) > 2 else ('adam_kernel' if any('adam_kernel' in n for n in names) else 'loss_finalize')
idx = [i for i, n in enumerate(names) if marker in n]
if len(idx) < 3:
    print('marker %r seen %d times: cannot cut steps' % (marker, len(idx)))
    sys.exit(0)
a, b = idx[-3], idx[-2]            # the last complete step but one (the very last may be the instrumented pass's)
step = names[a + 1:b + 1]
print('%d launches per step (between launches %d and %d of %s; %d steps in the trace)' % (len(step), len(idx) - 2, len(idx) - 1, marker, len(idx)))
cnt = Counter(re.sub(r'\(.*', '', n).replace('void pivp::', '').replace('pivp::', '')[:70] for n in step)
for n, c in cnt.most_common():
    print('  %5d  %s' % (c, n))
