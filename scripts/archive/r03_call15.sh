#!/bin/bash
# round 3, GPU call 15: norm-statistics finalize without per-partial divisions; fresh kernel breakdowns of both steps
O=$GRAFT_REPO_ROOT/gpurun_out/r03c15
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_conv_stats.py tests/test_resnext_hip.py tests/test_generator_module.py tests/test_mobilenet_train_hip.py tests/test_full_size_parity.py -m gpu -q --maxfail=20 > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|error" $O/tests.log | tail -2
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-300
R=r03
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_meta -o ${R} -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-drive > $O/prof_meta.log 2>&1
python scripts/step_breakdown.py $O/prof_meta/${R}_kernel_trace.csv > $O/step_breakdown_metatrain.csv 2>> $O/prof_meta.log
rm -rf $O/prof_meta
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ft -o ${R} -- python bench.py --workload finetune_step --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-drive > $O/prof_ft.log 2>&1
python scripts/step_breakdown.py $O/prof_ft/${R}_kernel_trace.csv > $O/step_breakdown_finetune.csv 2>> $O/prof_ft.log
rm -rf $O/prof_ft
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_metatrain.json 2> $O/bench_metatrain.err
python -c "
import json
j=json.load(open('$O/bench_metatrain.json')); print(j['value'], j['ms_per_step'])"
head -50 $O/step_breakdown_metatrain.csv | cut -c1-150
