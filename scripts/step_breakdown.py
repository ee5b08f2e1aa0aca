"""Per-step kernel breakdown from a rocprofv3 kernel trace of bench.py: picks one steady-state hipGraph replay of the
fine-tuning step (delimited by the mt_step_inc_kernel launches: one per optimizer step, G then D) and prints kernel
family, launches, total ms -- plus the GPU idle time inside the step.
usage: python scripts/step_breakdown.py <kernel_trace.csv> > breakdown.csv"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for x in csv.DictReader(f):
        rows.append((int(x['Start_Timestamp']), int(x['End_Timestamp']), x['Kernel_Name']))
rows.sort()
marks = [i for i, r in enumerate(rows) if 'mt_step_inc' in r[2]]
if len(marks) < 14:
    sys.exit('not enough optimizer steps in the trace')
# marks come in (G, D) pairs; a step = from one D-step marker to the next; take the 4th-from-last full step (inside the timed replays)
a, b = marks[-9], marks[-7]
seg = rows[a:b]
span = (seg[-1][0] - seg[0][0]) / 1e6
busy = sum(e - s for s, e, _ in seg[:-1]) / 1e6
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in seg[:-1]:
    k = n.split('(')[0]
    k = k.replace('void ', '')[:100]
    agg[k][0] += 1
    agg[k][1] += (e - s) / 1e6
print(f'# one steady-state step: span {span:.3f} ms, sum of kernel durations {busy:.3f} ms, kernels {len(seg) - 1}')
print('kernel,launches,total_ms,share')
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'"{k}",{c},{t:.4f},{t / busy:.4f}')
