"""From a rocprofv3 kernel trace (csv): wall time covered by kernels, sum of kernel durations, time with >= 2 kernels in flight."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for a, b, n, q in rows:
    ev.append((a, 1)); ev.append((b, -1))
ev.sort()
busy = over = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d; last = t
tot = sum(b - a for a, b, _, _ in rows)
print('span %.2f ms, covered by >=1 kernel %.2f ms, >=2 kernels %.2f ms, sum of durations %.2f ms, queues %s' % (
    (t1 - t0) / 1e6, busy / 1e6, over / 1e6, tot / 1e6, sorted(set(r[3] for r in rows))))
