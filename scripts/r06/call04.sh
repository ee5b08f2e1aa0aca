#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/c04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive"
for from in 3 2; do
LP_D_DPASS_FROM=$from LP_PARITY_OUT=$O timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "discriminator and f16" 2>&1 | grep -E "parity-256|passed|failed" | cut -c1-600 | tee -a $O/summary.txt
mv $O/r06_parity_gradients_f16.json $O/d_f16_from$from.json
LP_D_DPASS_FROM=$from timeout 300 $B > $O/bench_from$from.json 2> $O/bench_from$from.err
done
LP_PREC_G=f16 LP_D_DPASS_PREC=f16 timeout 300 $B > $O/bench_r05assign.json 2> $O/bench_r05assign.err
for f in $O/bench_*.json; do echo $f; python - $f <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))
except Exception as e: print('ERR',e)
P
done | tee -a $O/summary.txt
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o r06 -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-also --no-drive > $O/prof.log 2>&1
ls -la $O/prof/* | head; python scripts/step_breakdown.py $O/prof/*/r06_kernel_trace.csv > $O/step_breakdown.csv 2>> $O/prof.log || python scripts/step_breakdown.py $O/prof/r06_kernel_trace.csv > $O/step_breakdown.csv
head -3 $O/step_breakdown.csv
for f in $O/*.err; do echo $f; tail -2 $f | cut -c1-300; done
