#!/bin/bash
# round 3, GPU call 5: conv-epilogue statistics (+ the paths that now use them), prefetch diagnosis, meta-training + fine-tuning bench, breakdown
O=$GRAFT_REPO_ROOT/gpurun_out/r03c5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_stats.py tests/test_generator_module.py tests/test_resnext_hip.py tests/test_mobilenet_train_hip.py tests/test_metatrain_step.py tests/test_train_step.py tests/test_full_size_parity.py tests/test_fsth_plus.py -m gpu -q -s --maxfail=80 > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt
timeout 300 python scripts/prefetch_diag.py > $O/prefetch_diag.txt 2>&1
timeout 600 python bench.py --workload metatrain_step --steps 20 --warmup 5 --no-cpu-baseline --no-also > $O/bench_metatrain_f16.json 2> $O/bench_metatrain_f16.err
echo "bench rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --workload finetune_step --steps 20 --warmup 5 --no-cpu-baseline --no-also > $O/bench_finetune_f16.json 2> $O/bench_finetune_f16.err
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_meta -o r03 -- python bench.py --workload metatrain_step --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/prof_meta.log 2>&1
python scripts/step_breakdown.py $O/prof_meta/r03_kernel_trace.csv > $O/r03_step_breakdown_metatrain_f16.csv 2>> $O/prof_meta.log
python scripts/ktrace.py $O/prof_meta > $O/ktrace_meta.txt 2>&1
rm -f $O/prof_meta/r03_kernel_trace.csv
grep -E "passed|failed|error" $O/tests.log | tail -3
grep -E "FAILED|\[parity\] meta-train step, train" $O/tests.log | cut -c1-420
cat $O/prefetch_diag.txt | tail -12
cut -c1-260 $O/bench_metatrain_f16.json
cut -c1-260 $O/bench_finetune_f16.json
tail -2 $O/bench_metatrain_f16.err
head -40 $O/r03_step_breakdown_metatrain_f16.csv
