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
if len(marks) < 26:
    sys.exit('not enough optimizer steps in the trace')
# a step holds 2 markers (G, D) -- 3 when the generator's slice of optimizer_G is stepped on its own (round 6: the early updates beside the
# encoders' backward) -- so the period is read off the trace: the smallest p for which the kernel counts between consecutive markers repeat
# with period p over the last replays.  A step = from one marker to the p-th next; take the 4th-from-last full step (inside the timed replays)
marks = marks[:-6]          # (bench.py ends with two EAGER instrumented steps -- the live roofline figure: up to 6 markers that are not graph replays)
gaps = [marks[i + 1] - marks[i] for i in range(len(marks) - 1)]
def close(x, y):          # (kernels of concurrent streams interleave a little differently from replay to replay)
    return abs(x - y) <= max(4, 0.03 * max(x, y))
per = next((p for p in (1, 2, 3, 4, 5, 6) if len(gaps) >= 4 * p and all(close(gaps[-1 - i], gaps[-1 - i - p]) for i in range(3 * p))), 2)
print(f'markers {len(marks)}, last gaps {gaps[-12:]}, period {per}', file=sys.stderr)
a, b = marks[-1 - 4 * per], marks[-1 - 3 * per]
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
