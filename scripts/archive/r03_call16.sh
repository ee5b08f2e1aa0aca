#!/bin/bash
# round 3, GPU call 16: independent branches of the step on separate streams (pose encoder || identity encoder, VGG criterions || discriminator)
O=$GRAFT_REPO_ROOT/gpurun_out/r03c16
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_metatrain_step.py tests/test_train_step.py tests/test_train_entry_gpu.py tests/test_data_parallel_gpu.py tests/test_checkpoint_fixture.py tests/test_prefetch.py -m gpu -q --maxfail=20 > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|error" $O/tests.log | tail -2
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-300
for v in 1 0; do
  LP_OVERLAP=$v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_metatrain_ov$v.json 2> $O/bench_metatrain_ov$v.err
  LP_OVERLAP=$v timeout 300 python bench.py --workload finetune_step --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_finetune_ov$v.json 2> $O/bench_finetune_ov$v.err
done
python -c "
import json
for f in ('bench_metatrain_ov1', 'bench_metatrain_ov0', 'bench_finetune_ov1', 'bench_finetune_ov0'):
    try:
        j=json.load(open('$O/%s.json' % f)); print(f, j['value'], j['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)"
tail -2 $O/bench_metatrain_ov1.err
