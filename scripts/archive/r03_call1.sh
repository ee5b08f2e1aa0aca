#!/bin/bash
# round 3, GPU call 1: parity of the new ResNeXt-50 HIP path, the existing suite, the meta-training bench line + one-step breakdown
O=$GRAFT_REPO_ROOT/gpurun_out/r03c1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_resnext_hip.py -m gpu -q -s --maxfail=80 > $O/resnext_tests.log 2>&1
echo "resnext tests rc=$?" | tee -a $O/summary.txt
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_resnext_hip.py --maxfail=20 > $O/all_tests.log 2>&1
echo "other tests rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --workload metatrain_step --steps 20 --warmup 5 --no-cpu-baseline --no-also > $O/bench_metatrain_f16.json 2> $O/bench_metatrain_f16.err
echo "bench rc=$?" | tee -a $O/summary.txt
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_meta -o r03 -- python bench.py --workload metatrain_step --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/prof_meta.log 2>&1
python scripts/step_breakdown.py $O/prof_meta/r03_kernel_trace.csv > $O/r03_step_breakdown_metatrain_f16.csv 2>> $O/prof_meta.log
rm -f $O/prof_meta/r03_kernel_trace.csv
grep -E "passed|failed|error" $O/resnext_tests.log | tail -3
grep -E "^\[parity\] resnext50" $O/resnext_tests.log
grep -E "passed|failed|error" $O/all_tests.log | tail -3
cut -c1-600 $O/bench_metatrain_f16.json
tail -3 $O/bench_metatrain_f16.err
head -40 $O/r03_step_breakdown_metatrain_f16.csv
