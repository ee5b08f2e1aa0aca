#!/bin/bash
# round 3, GPU call 22: the identity encoder's weight gradients on a side stream -- parity + A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r03c22
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_resnext_hip.py tests/test_metatrain_step.py tests/test_data_parallel_gpu.py -m gpu -q --maxfail=20 > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|error" $O/tests.log | tail -2
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-300
{
for v in 1 0 1 0; do
LP_OVERLAP_WGRAD=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('metatrain wgrad side stream=$v', j['value'], j['ms_per_step'])"
done
} 2>&1 | tee $O/r03_wgrad_stream.txt
