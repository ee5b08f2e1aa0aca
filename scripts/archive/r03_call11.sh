#!/bin/bash
# round 3, GPU call 11: BatchNorm-side kernels -- adaptive stage-1 split and the rows-form apply (A/B), affected parity tests, step bench
O=$GRAFT_REPO_ROOT/gpurun_out/r03c11
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LP_STAT_SPLIT_FIXED=1 LP_BNB_ITEMS=1 timeout 200 python scripts/bn_micro.py > $O/bn_micro_old.txt 2>&1
timeout 200 python scripts/bn_micro.py > $O/bn_micro_new.txt 2>&1
echo "== fixed 1024-pixel split, one item per thread (rounds 1-2 / call 10)"; cat $O/bn_micro_old.txt | grep -v amdgpu.ids
echo "== adaptive split, rows-form apply"; cat $O/bn_micro_new.txt | grep -v amdgpu.ids
timeout 1500 python -m pytest tests/test_hip_ops.py tests/test_conv_stats.py tests/test_resnext_hip.py tests/test_mobilenet_train_hip.py tests/test_generator_module.py tests/test_mobilenet_hip.py tests/test_full_size_parity.py tests/test_metatrain_step.py -m gpu -q --maxfail=80 > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_metatrain.json 2> $O/bench_metatrain.err
LP_STAT_SPLIT_FIXED=1 LP_BNB_ITEMS=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_metatrain_old.json 2> $O/bench_metatrain_old.err
timeout 300 python bench.py --workload finetune_step --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_finetune.json 2> $O/bench_finetune.err
grep -E "passed|failed|error" $O/tests.log | tail -3
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-300
python -c "
import json
for f in ('bench_metatrain', 'bench_metatrain_old', 'bench_finetune'):
    try:
        j=json.load(open('$O/%s.json' % f)); print(f, j['value'], j['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)"
