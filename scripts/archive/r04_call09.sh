#!/bin/bash
# round 4, GPU call 9: the chunk-pipelined 1x1 kernel: parity cases, micro A/B (fp16 and bf16x3), step time
O=$GRAFT_REPO_ROOT/gpurun_out/r04c09
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_conv_pipe.py -m gpu -q -x -s > $O/pipe_tests.log 2>&1; echo "pipe tests rc=$?" | tee $O/summary.txt
grep -E "passed|failed|FAILED" $O/pipe_tests.log | tail -5
for pr in 2 1; do for v in "LP_CONV1X1_PIPE=0" "LP_CONV1X1_PIPE=1" "LP_CONV1X1_PIPE=1 LP_CONV1X1_PIPE_MAXK=4096"; do
  echo "== PREC=$pr $v" >> $O/micro.log
  env $v SHAPES=1x1 PREC=$pr WHAT=conv REPS=30 timeout 120 python scripts/conv_micro.py 2>&1 | grep -v amdgpu >> $O/micro.log
done; done
cat $O/micro.log
for v in 0 1; do
  LP_CONV1X1_PIPE=$v timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_p$v.json 2> $O/bench_p$v.err
  python - <<PY
import json
try:
    j=json.load(open('$O/bench_p$v.json')); print('bench LP_CONV1X1_PIPE=$v', j['ms_per_step'], 'ms', j['value'], 'img/s', 'conv1x1', j['roofline_conv1x1']['achieved'], 'GB/s')
except Exception as e: print('bench $v failed', e)
PY
done
