#!/bin/bash
# round 4, GPU call 2: identity-encoder precision sweep (bf16x3 head + fp16 tail of K blocks) at the full geometry; conv_pipe default-heuristic tests
O=$GRAFT_REPO_ROOT/gpurun_out/r04c02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 0 3 5 7 9 12; do
  LP_E_F16_TAIL=$t timeout 120 python scripts/e1_parity_full.py 64 256 noise 2>&1 | grep -E "e1-parity|Error|error" | tail -2 >> $O/e1_sweep.log
done
LP_PREC_E=f16 timeout 120 python scripts/e1_parity_full.py 64 256 noise 2>&1 | grep -E "e1-parity|Error" | tail -2 >> $O/e1_sweep.log
LP_E_F16_TAIL=5 timeout 120 python scripts/e1_parity_full.py 64 256 smooth 2>&1 | grep -E "e1-parity|Error" | tail -2 >> $O/e1_sweep.log
cat $O/e1_sweep.log
timeout 300 python -m pytest tests/test_conv_pipe.py tests/test_resnext_hip.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt
grep -E "passed|failed|FAILED" $O/tests.log | tail -5
for t in 3 6; do
  LP_E_F16_TAIL=$t timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_tail$t.json 2> $O/bench_tail$t.err
  python - <<PY
import json
try:
    j=json.load(open('$O/bench_tail$t.json')); print('bench tail=$t', j['ms_per_step'], 'ms', j['value'], 'img/s', 'roof', j['roofline']['frac'])
except Exception as e: print('bench tail $t failed', e)
PY
done
