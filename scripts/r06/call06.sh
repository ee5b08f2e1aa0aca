#!/bin/bash
# round 6, call 6: tie-masked full-depth identity-encoder gradient test; E tweaks (mask from planes); FETCH_SIZE A/B of the XCD order
O=$GRAFT_REPO_ROOT/gpurun_out/c06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LP_PARITY_OUT=$O timeout 1500 python -m pytest tests/test_e1_full_gpu.py tests/test_resnext_hip.py -m gpu -q -s 2>&1 | grep -E "^\[e1|passed|failed|^FAILED|Error|^E  " | cut -c1-1200 > $O/parity.txt
tail -25 $O/parity.txt | cut -c1-700
for x in 0 1; do
  LP_CONV_XCD=$x PREC=2 WHAT=conv REPS=4 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "conv_pipe_kernel|conv_dma_kernel" --output-format csv -d $O/pmc_f$x -o p -- python scripts/conv_micro.py > $O/pmc_f$x.log 2>&1
  python scripts/pmc_summary.py $O/pmc_f$x/p_counter_collection.csv > $O/pmc_fetch_xcd$x.csv 2> $O/pmc_sum$x.err
  LP_CONV_XCD=$x PREC=1 WHAT=conv REPS=4 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "conv_pipe_kernel|conv_dma_kernel" --output-format csv -d $O/pmc_g$x -o p -- python scripts/conv_micro.py > $O/pmc_g$x.log 2>&1
  python scripts/pmc_summary.py $O/pmc_g$x/p_counter_collection.csv > $O/pmc_fetch_x3_xcd$x.csv 2>> $O/pmc_sum$x.err
  rm -rf $O/pmc_f$x $O/pmc_g$x
done
echo "== f16 FETCH xcd0 | xcd1"; paste -d'|' <(cut -d, -f1,3,5 $O/pmc_fetch_xcd0.csv | cut -c1-110) <(cut -d, -f3,5 $O/pmc_fetch_xcd1.csv) | head -16
echo "== bf16x3 FETCH xcd0 | xcd1"; paste -d'|' <(cut -d, -f1,3,5 $O/pmc_fetch_x3_xcd0.csv | cut -c1-110) <(cut -d, -f3,5 $O/pmc_fetch_x3_xcd1.csv) | head -16
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive"
LP_E_MASK16=0 timeout 300 $B > $O/bench_mask0.json 2> $O/bench_mask0.err
timeout 300 $B > $O/bench_mask1.json 2> $O/bench_mask1.err
for f in $O/bench_*.json; do echo $f; python - $f <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))
except Exception as e: print('ERR',e)
P
done
for f in $O/*.err; do tail -1 $f | grep -v amdgpu.ids | cut -c1-300; done
