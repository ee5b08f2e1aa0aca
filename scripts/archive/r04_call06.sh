#!/bin/bash
# round 4, GPU call 6: encoders' backward beside loss_D.backward (A/B), train-step goldens with the cut graph, parity tests on the final gates
O=$GRAFT_REPO_ROOT/gpurun_out/r04c06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_metatrain_step.py tests/test_streams_gpu.py tests/test_train_step.py -m gpu -q -x > $O/tests.log 2>&1; echo "step tests rc=$?" | tee $O/summary.txt
grep -E "passed|failed|FAILED|Error" $O/tests.log | tail -5
for v in 0 1; do
  LP_OVERLAP_EBWD=$v timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_ebwd$v.json 2> $O/bench_ebwd$v.err
  python - <<PY
import json
try:
    j=json.load(open('$O/bench_ebwd$v.json')); print('bench LP_OVERLAP_EBWD=$v', j['ms_per_step'], 'ms', j['value'], 'img/s', 'roof', j['roofline']['frac'])
except Exception as e: print('bench ebwd $v failed', e)
PY
  tail -2 $O/bench_ebwd$v.err | cut -c1-300
done
LP_PARITY_OUT=$O timeout 600 python -m pytest tests/test_metatrain_full_gpu.py -m gpu -q -s > $O/full.log 2>&1; echo "full configs2 rc=$?" | tee -a $O/summary.txt
grep -E "parity-configs2|passed|failed|Error" $O/full.log | cut -c1-1500 | tail -6
