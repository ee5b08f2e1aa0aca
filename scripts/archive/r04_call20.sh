#!/bin/bash
# round 4, GPU call 20: wgrad3_pipe -- kernel vs reduction time per layer class (kernel trace), fewer pixel splits (LP_WGRAD3_WGS = 256 / 128)
O=$GRAFT_REPO_ROOT/gpurun_out/r04c20
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "LP_WGRAD3_WGS=256" "LP_WGRAD3_WGS=128"; do
  echo "== $v" >> $O/micro.log
  env $v SHAPES=wgrad BIAS=1 PREC=2 WHAT=wgrad REPS=30 timeout 120 python scripts/conv_micro.py 2>&1 | grep -v amdgpu.ids >> $O/micro.log
done
cat $O/micro.log
for v in "LP_WGRAD3_WGS=1024" "LP_WGRAD3_WGS=256"; do
  rm -rf $O/tr
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- env $v SHAPES=wgrad BIAS=1 PREC=2 WHAT=wgrad REPS=10 python scripts/conv_micro.py > $O/tr.log 2>&1
  echo "== trace $v" | tee -a $O/trace_summary.txt
  python - <<PY | tee -a $O/trace_summary.txt
import csv, glob, collections
f = glob.glob('$O/tr/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'wgrad' not in k: continue
    key = (k.split('(')[0][:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'))
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for (k, g), (n, us) in agg.items():
    print(f'{k:62s} grid {g:>8s} x{n:3d} avg {us / n:7.1f} us')
PY
done
rm -rf $O/tr
