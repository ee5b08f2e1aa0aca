"""Concurrency picture of one steady-state training step from a rocprofv3 kernel trace (the same step cut as scripts/step_breakdown.py):
how much of the step's wall time has 0 / 1 / 2 / >= 3 kernels in flight, per-queue busy time, and -- per kernel family -- its EXCLUSIVE time
(it is the only kernel running: the closest a kernel trace gets to "on the critical path") beside its total time.
usage: python scripts/trace_concurrency.py <kernel_trace.csv> [top_n] > concurrency.csv"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for x in csv.DictReader(f):
        rows.append((int(x['Start_Timestamp']), int(x['End_Timestamp']), x['Kernel_Name'], x.get('Queue_Id', '?'), x.get('Stream_Id', '?')))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
rows.sort()
marks = [i for i, r in enumerate(rows) if 'mt_step_inc' in r[2]]
if len(marks) < 14:
    sys.exit('not enough optimizer steps in the trace')
a, b = marks[-9], marks[-7]          # the step scripts/step_breakdown.py cuts (the last two steps of a bench.py trace are its eager one-stream instrumented steps)
seg = rows[a:b]
t0, t1 = seg[0][0], max(e for _, e, *_ in seg)
ev = []
for i, (s, e, n, q, st) in enumerate(seg):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
live = set()
hist = collections.Counter()
excl = collections.defaultdict(float)
tot = collections.defaultdict(lambda: [0, 0.0])
qbusy = collections.defaultdict(float)
prev = t0
for t, d, i in ev:
    dt = t - prev
    if dt > 0:
        hist[min(len(live), 3)] += dt
        if len(live) == 1:
            (j,) = live
            excl[seg[j][2].split('(')[0].replace('void ', '')[:90]] += dt
    prev = t
    if d > 0:
        live.add(i)
    else:
        live.discard(i)
for s, e, n, q, st in seg:
    k = n.split('(')[0].replace('void ', '')[:90]
    tot[k][0] += 1; tot[k][1] += e - s
    qbusy[(q, st)] += e - s
span = (t1 - t0) / 1e6
print(f'# step span {span:.3f} ms; kernels {len(seg)}; wall time with 0 / 1 / 2 / >=3 kernels in flight: '
      + ' / '.join(f'{hist[k] / 1e6:.2f}' for k in range(4)) + ' ms')
print('# busy ms per (queue, stream): ' + ', '.join(f'{k}: {v / 1e6:.2f}' for k, v in sorted(qbusy.items(), key=lambda kv: -kv[1])[:12]))
print('kernel,launches,total_ms,exclusive_ms')
for k, v in sorted(excl.items(), key=lambda kv: -kv[1])[:top]:
    print(f'"{k}",{tot[k][0]},{tot[k][1] / 1e6:.3f},{v / 1e6:.3f}')
