#!/bin/bash
# round 4, GPU call 16: wgrad1x1 with 32-pixel stages (bf16x3: two workgroups per CU instead of one; fp16: four instead of two): parity, micro, step
O=$GRAFT_REPO_ROOT/gpurun_out/r04c16
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_resnext_hip.py -m gpu -q -x -k "pointwise_weight_gradient" > $O/tests.log 2>&1; echo "wgrad1x1 tests rc=$?"; tail -2 $O/tests.log
LP_W1_KP=32 timeout 300 python -m pytest tests/test_resnext_hip.py -m gpu -q -x -k "pointwise_weight_gradient" > $O/tests32.log 2>&1; echo "wgrad1x1 tests KP=32 all modes rc=$?"; tail -2 $O/tests32.log
for pr in 1 2; do for v in "LP_W1_KP=64" "LP_W1_KP=32"; do
  echo "== PREC=$pr $v" >> $O/micro.log
  env $v SHAPES=1x1 PREC=$pr WHAT=wgrad REPS=30 timeout 120 python scripts/conv_micro.py 2>&1 | grep -v amdgpu >> $O/micro.log
done; done
cat $O/micro.log
for v in "LP_W1_KP=64" "A=1" "LP_W1_KP=32" "A=1"; do
  env $v timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/b.json 2> $O/b.err
  echo "$v"; python -c "
import json; j=json.load(open('$O/b.json')); print('   ', j['ms_per_step'], 'ms', j['value'], 'img/s')"
done
