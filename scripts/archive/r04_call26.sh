#!/bin/bash
# round 4, GPU call 26: grouped weight gradient on the diagonal tiles only (wgrad3_pipe DIAGB = 16 / 32, all three precision modes): parity + step A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r04c26
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_wgrad3_pipe.py tests/test_resnext_hip.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt
grep -E "passed|failed|FAILED|Error|grouped\] prec=1" $O/tests.log | cut -c1-300 | tail -14
for v in "LP_GWGRAD_DIAG=0" "LP_GWGRAD_DIAG=1" "LP_GWGRAD_DIAG=0" "LP_GWGRAD_DIAG=1"; do
  env $v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-also --no-drive 2>$O/b.err | python -c "import json,sys; d=json.load(sys.stdin); print('$v', d['ms_per_step'], 'ms', d.get('roofline_gconv_wgrad', {}).get('avg_launch_us'), 'us per gconv wgrad launch')" | tee -a $O/summary.txt
done
