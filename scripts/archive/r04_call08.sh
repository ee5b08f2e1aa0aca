#!/bin/bash
# round 4, GPU call 8: A/B of the encoders' backward beside loss_D.backward on the captured full-size step
O=$GRAFT_REPO_ROOT/gpurun_out/r04c08
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  LP_OVERLAP_EBWD=$v timeout 200 python -X faulthandler bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_ebwd$v.json 2> $O/bench_ebwd$v.err
  python - <<PY
import json
try:
    j=json.load(open('$O/bench_ebwd$v.json')); print('bench LP_OVERLAP_EBWD=$v', j['ms_per_step'], 'ms', j['value'], 'img/s')
except Exception as e: print('bench ebwd $v failed', e)
PY
  tail -12 $O/bench_ebwd$v.err | grep -v amdgpu.ids | cut -c1-300
done
