#!/bin/bash
# round 3, GPU call 31: the discriminator's three passes beside each other (forward, and through autograd the whole loss_D.backward): parity + A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r03c31
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
for v in 1 0 1 0; do
LP_OVERLAP_DPASSES=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('metatrain, three D passes beside each other=$v', j['value'], j['ms_per_step'])"
done
for v in 1 0; do
LP_OVERLAP_DPASSES=$v timeout 300 python bench.py --workload finetune_step --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('finetune, three D passes beside each other=$v', j['value'], j['ms_per_step'])"
done
} 2>&1 | tee $O/r03_dpasses.txt
LP_OVERLAP_DPASSES=1 timeout 600 python -m pytest tests/test_streams_gpu.py tests/test_metatrain_step.py tests/test_discriminator_criterions.py tests/test_data_parallel_gpu.py -m gpu -q -s > $O/tests.log 2>&1
echo "tests rc=$?" | tee $O/summary.txt
grep -E "\[streams\]|passed|failed" $O/tests.log | cut -c1-300 | tail -5
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-300
