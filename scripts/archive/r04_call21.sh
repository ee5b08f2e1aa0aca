#!/bin/bash
# round 4, GPU call 21: wgrad3_pipe + slab-group flat reduction: parity (kernel cases, per-op tests incl. spectral-norm dot, critic / generator module tests), micro, step A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r04c21
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_wgrad3_pipe.py tests/test_hip_ops.py tests/test_discriminator_criterions.py tests/test_generator_module.py tests/test_resnext_hip.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt
grep -E "passed|failed|FAILED" $O/tests.log | cut -c1-220 | tail -10
echo "== LP_WGRAD3_PIPE=1" >> $O/micro.log
SHAPES=wgrad BIAS=1 PREC=2 WHAT=wgrad REPS=30 timeout 120 python scripts/conv_micro.py 2>&1 | grep -v amdgpu.ids >> $O/micro.log
cat $O/micro.log
for v in "LP_WGRAD3_PIPE=0" "LP_WGRAD3_PIPE=1" "LP_WGRAD3_PIPE=0" "LP_WGRAD3_PIPE=1"; do
  env $v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-also --no-drive > $O/b.json 2> $O/b.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open('$O/b.json')); print('$v', d['ms_per_step'], 'ms', d['value'], 'img/s', 'wgrad frac', d.get('roofline_conv_wgrad', {}).get('frac'))
except Exception as e:
    print('$v bench failed', e, open('$O/b.err').read()[-1500:])
PY
done
