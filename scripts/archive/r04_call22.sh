#!/bin/bash
# round 4, GPU call 22: target-image work beside the GENERATOR forward only: VGG target halves (LP_OVERLAP_TARGETS=2), the critic's real pass (LP_OVERLAP_REAL=1)
O=$GRAFT_REPO_ROOT/gpurun_out/r04c22
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "X=0" "LP_OVERLAP_TARGETS=2" "LP_OVERLAP_REAL=1" "LP_OVERLAP_TARGETS=2 LP_OVERLAP_REAL=1" "X=0" "LP_OVERLAP_REAL=1"; do
  env $v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-also --no-drive > $O/b.json 2> $O/b.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open('$O/b.json')); print('$v', d['ms_per_step'], 'ms', d['value'], 'img/s')
except Exception as e:
    print('$v bench failed', e, open('$O/b.err').read()[-2500:])
PY
done
