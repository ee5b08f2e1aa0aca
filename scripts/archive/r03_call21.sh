#!/bin/bash
# round 3, GPU call 21: target-image VGG features ahead of the generator, optimizer_G.step + EMA beside the D backward; parity + A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r03c21
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_metatrain_step.py tests/test_train_step.py tests/test_train_entry_gpu.py tests/test_data_parallel_gpu.py tests/test_checkpoint_fixture.py tests/test_prefetch.py tests/test_discriminator_criterions.py -m gpu -q --maxfail=20 > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|error" $O/tests.log | tail -2
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-300
{
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('metatrain default', j['value'], j['ms_per_step'])"
LP_OVERLAP_OPTIMIZER=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('metatrain optimizer=0', j['value'], j['ms_per_step'])"
timeout 300 python bench.py --workload finetune_step --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('finetune default', j['value'], j['ms_per_step'])"
LP_OVERLAP_OPTIMIZER=0 timeout 300 python bench.py --workload finetune_step --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('finetune optimizer=0', j['value'], j['ms_per_step'])"
timeout 300 python scripts/prefetch_overlap_diag.py metatrain 2>&1 | grep "input path"
} 2>&1 | tee $O/r03_stream_overlap2.txt
