#!/bin/bash
# round 4, GPU call 25: grouped conv forward / dgrad skipping the zero half of each 64-channel block (LP_GCONV_HALF): parity tests + step A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r04c25
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_resnext_hip.py tests/test_conv_stats.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt
grep -E "passed|failed|FAILED|Error" $O/tests.log | cut -c1-300 | tail -6
for v in "LP_GCONV_HALF=0" "LP_GCONV_HALF=1" "LP_GCONV_HALF=0" "LP_GCONV_HALF=1"; do
  env $v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-also --no-drive 2>$O/b.err | python -c "import json,sys; d=json.load(sys.stdin); print('$v', d['ms_per_step'], 'ms', d.get('roofline_gconv', {}).get('avg_launch_us'), 'us per gconv launch')" | tee -a $O/summary.txt
done
