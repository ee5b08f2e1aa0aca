#!/bin/bash
# round 4, GPU call 1: first run of the tap-pipelined conv kernel (parity + micro A/B), identity-encoder parity at the full geometry,
# bench self-launch test, step time with the identity encoder in bf16x3
O=$GRAFT_REPO_ROOT/gpurun_out/r04c01
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_conv_pipe.py -m gpu -q -x -s > $O/pipe_tests.log 2>&1; echo "pipe tests rc=$?" | tee $O/summary.txt
grep -E "passed|failed|FAILED" $O/pipe_tests.log | tail -5
for v in "LP_CONV_PIPE=0" "LP_CONV_PIPE=1" "LP_CONV_PIPE_MR=8" "LP_CONV_PIPE_MR=4"; do
  echo "== $v" >> $O/micro.log
  env $v PREC=2 WHAT=conv REPS=30 timeout 120 python scripts/conv_micro.py 2>&1 | grep -v amdgpu.ids >> $O/micro.log
done
cat $O/micro.log
for c in noise smooth; do for pe in f16 bf16x3; do
  LP_PREC_E=$pe timeout 240 python scripts/e1_parity_full.py 64 256 $c 2>&1 | grep e1-parity >> $O/e1_parity.log
done; done
cat $O/e1_parity.log
timeout 400 python -m pytest tests/test_data_parallel_gpu.py -m gpu -q -x -k "bench_starts" > $O/dp_bench.log 2>&1; echo "dp bench test rc=$?" | tee -a $O/summary.txt
tail -5 $O/dp_bench.log
for pe in f16 bf16x3; do
  LP_PREC_E=$pe timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_E_$pe.json 2> $O/bench_E_$pe.err
  python - <<PY
import json
try:
    j=json.load(open('$O/bench_E_$pe.json')); print('bench LP_PREC_E=$pe', j['ms_per_step'], 'ms', j['value'], 'img/s', 'roof', j['roofline']['frac'])
except Exception as e: print('bench $pe failed', e)
PY
done
