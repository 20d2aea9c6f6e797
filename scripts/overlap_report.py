"""From a rocprofv3 kernel trace (csv): wall time covered by kernels, sum of kernel durations, time with >= 2 kernels in flight,
and per queue: busy time, and the time it sat idle while the OTHER queue was busy (= it was waiting for an event, or had no work).
python scripts/overlap_report.py <kernel_trace.csv> [skip_fraction]   (skip the first part of the trace: warm-up, default 0.3)"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
T0, T1 = rows[0][0], max(r[1] for r in rows)
cut = T0 + (T1 - T0) * skip
rows = [r for r in rows if r[0] >= cut]
t0, t1 = rows[0][0], max(r[1] for r in rows)


def covered(rs):
    ev = []
    for a, b, _, _ in rs:
        ev.append((a, 1)); ev.append((b, -1))
    ev.sort()
    busy = over = 0; depth = 0; last = ev[0][0]
    segs = []
    for t, d in ev:
        if depth >= 1:
            busy += t - last
            if segs and segs[-1][1] == last:
                segs[-1][1] = t
            else:
                segs.append([last, t])
        if depth >= 2:
            over += t - last
        depth += d; last = t
    return busy, over, segs


busy, over, _ = covered(rows)
tot = sum(b - a for a, b, _, _ in rows)
queues = sorted(set(r[3] for r in rows))
print('span %.2f ms, covered by >=1 kernel %.2f ms, >=2 kernels %.2f ms, sum of durations %.2f ms, queues %s' % (
    (t1 - t0) / 1e6, busy / 1e6, over / 1e6, tot / 1e6, queues))
per = {q: covered([r for r in rows if r[3] == q]) for q in queues}


def inter(sa, sb):      # total overlap of two sorted interval lists
    i = j = 0; s = 0
    while i < len(sa) and j < len(sb):
        lo, hi = max(sa[i][0], sb[j][0]), min(sa[i][1], sb[j][1])
        if hi > lo:
            s += hi - lo
        if sa[i][1] < sb[j][1]:
            i += 1
        else:
            j += 1
    return s


for q in queues:
    b, _, segs = per[q]
    others = [r for r in rows if r[3] != q]
    line = 'queue %s: busy %.2f ms' % (q, b / 1e6)
    if others:
        ob, _, osegs = covered(others)
        line += ', idle while another queue runs %.2f ms' % ((ob - inter(segs, osegs)) / 1e6)
    print(line)
# the kernels after which a queue's longest idle stretches begin (what was it waiting for?)
for q in queues:
    rs = [r for r in rows if r[3] == q]
    gaps = defaultdict(lambda: [0, 0.0])
    for (a0, b0, n0, _), (a1, b1, n1, _) in zip(rs[:-1], rs[1:]):
        g = a1 - b0
        if g > 20000:
            k = (n0.split('(')[0][-40:], n1.split('(')[0][-40:])
            gaps[k][0] += 1; gaps[k][1] += g / 1e3
    top = sorted(gaps.items(), key=lambda kv: -kv[1][1])[:8]
    for (n0, n1), (cnt, us) in top:
        print('  queue %s: %4d gaps > 20 us, %8.1f us in all, between %s -> %s' % (q, cnt, us, n0, n1))
