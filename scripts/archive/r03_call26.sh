#!/bin/bash
# round 3, GPU call 26: criterion overlap restored (target_rgbs lives in data_dict); target features ahead of the generator: real A/B; parity
O=$GRAFT_REPO_ROOT/gpurun_out/r03c26
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
for v in 1 0 1 0; do
LP_OVERLAP_TARGETS=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('metatrain, criterions on side streams, target features ahead=$v', j['value'], j['ms_per_step'])"
done
for v in 1 0; do
LP_OVERLAP_TARGETS=$v timeout 300 python bench.py --workload finetune_step --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('finetune target features ahead=$v', j['value'], j['ms_per_step'])"
done
} 2>&1 | tee $O/r03_targets_ahead2.txt
timeout 900 python -m pytest tests/test_metatrain_step.py tests/test_train_step.py tests/test_train_entry_gpu.py tests/test_prefetch.py tests/test_data_parallel_gpu.py -m gpu -q --maxfail=20 > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|error" $O/tests.log | tail -2
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-300
