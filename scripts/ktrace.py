"""median duration of every (kernel, grid) pair of a rocprofv3 --kernel-trace CSV, in first-appearance order.
usage: python scripts/ktrace.py <rocprof output dir> [name filter]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
agg = {}
for r in csv.DictReader(open(f)):
    nm = r["Kernel_Name"]
    if flt not in nm:
        continue
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    key = (nm[:70], r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"), r.get("Workgroup_Size_X"))
    agg.setdefault(key, []).append(d)
for k, ds in agg.items():
    ds = sorted(ds)
    print(f'{k[0]:72s} grid {k[1]:>7s} x{k[2]:>3s} x{k[3]:>3s} wg {k[4]:>4s}  n {len(ds):4d}  median {ds[len(ds) // 2] / 1e3:8.2f} us')
