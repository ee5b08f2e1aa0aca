#!/bin/bash
# round 6, call 1: (a) XCD-ordered conv grids: parity + A/B; (b) what the 1e-3 gradient gate costs: generator strict, critic D-passes strict
O=$GRAFT_REPO_ROOT/gpurun_out/c01; mkdir -p $O
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive"
timeout 600 python -m pytest tests/test_conv_pipe.py tests/test_conv_stats.py tests/test_kernel_variants.py tests/test_resnext_hip.py -m gpu -q 2>&1 | tail -5 > $O/pytest_conv.txt
LP_CONV_XCD=0 timeout 300 $B > $O/bench_xcd0.json 2> $O/bench_xcd0.err
timeout 300 $B > $O/bench_xcd1.json 2> $O/bench_xcd1.err
LP_PREC_G=bf16x3 timeout 300 $B > $O/bench_gx3.json 2> $O/bench_gx3.err
LP_PREC_G=bf16x3 LP_D_DPASS_PREC=bf16x3 LP_D_DPASS_FROM=0 timeout 300 $B > $O/bench_gx3_dx3.json 2> $O/bench_gx3_dx3.err
LP_PREC_G=bf16x3 LP_D_DPASS_PREC=bf16x3 LP_D_DPASS_FROM=4 timeout 300 $B > $O/bench_gx3_dx3from4.json 2> $O/bench_gx3_dx3from4.err
for f in $O/bench_*.json; do echo $f; python - $f <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))
except Exception as e: print('ERR',e)
P
done | tee $O/summary.txt
LP_PARITY_OUT=$O timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "discriminator and f16" 2>&1 | grep -E "parity-256|passed|failed" | cut -c1-600 | tee -a $O/summary.txt
mv $O/r06_parity_gradients_f16.json $O/d_f16.json
LP_D_DPASS_PREC=bf16x3 LP_D_DPASS_FROM=4 LP_PARITY_OUT=$O timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "discriminator and f16" 2>&1 | grep -E "parity-256|passed|failed" | cut -c1-600 | tee -a $O/summary.txt
mv $O/r06_parity_gradients_f16.json $O/d_f16_dx3from4.json
LP_D_DPASS_PREC=bf16x3 LP_D_DPASS_FROM=0 LP_PARITY_OUT=$O timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "discriminator and f16" 2>&1 | grep -E "parity-256|passed|failed" | cut -c1-600 | tee -a $O/summary.txt
mv $O/r06_parity_gradients_f16.json $O/d_f16_dx3.json
for f in $O/*.err; do echo $f; tail -3 $f | cut -c1-300; done
cat $O/pytest_conv.txt
