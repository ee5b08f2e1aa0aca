#!/bin/bash
# round 3, GPU call 3: whole suite (new: MobileNetV2 training path, dice / hinge kernels, 1x1 wgrad with 128-channel workgroups),
# meta-training bench + one-step breakdown, A/B of the 1x1 weight-gradient workgroup size
O=$GRAFT_REPO_ROOT/gpurun_out/r03c3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_resnext_hip.py tests/test_mobilenet_train_hip.py tests/test_metatrain_step.py -m gpu -q -s --maxfail=80 > $O/embedder_tests.log 2>&1
echo "embedder tests rc=$?" | tee -a $O/summary.txt
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_resnext_hip.py --deselect tests/test_mobilenet_train_hip.py --deselect tests/test_metatrain_step.py --maxfail=30 > $O/all_tests.log 2>&1
echo "other tests rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --workload metatrain_step --steps 20 --warmup 5 --no-cpu-baseline --no-also > $O/bench_metatrain_f16.json 2> $O/bench_metatrain_f16.err
echo "bench rc=$?" | tee -a $O/summary.txt
LP_WGRAD_COB1=64 timeout 600 python bench.py --workload metatrain_step --steps 20 --warmup 5 --no-cpu-baseline --no-also > $O/bench_metatrain_f16_cob64.json 2> $O/bench_metatrain_f16_cob64.err
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_meta -o r03 -- python bench.py --workload metatrain_step --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/prof_meta.log 2>&1
python scripts/step_breakdown.py $O/prof_meta/r03_kernel_trace.csv > $O/r03_step_breakdown_metatrain_f16.csv 2>> $O/prof_meta.log
rm -f $O/prof_meta/r03_kernel_trace.csv
grep -E "passed|failed|error" $O/embedder_tests.log | tail -3
grep -E "^\[parity\] (shallow|resnext50|mobilenet_v2|meta-train)|FAILED" $O/embedder_tests.log | cut -c1-700
grep -E "passed|failed|error|FAILED" $O/all_tests.log | tail -8
cut -c1-300 $O/bench_metatrain_f16.json
cut -c1-300 $O/bench_metatrain_f16_cob64.json
tail -2 $O/bench_metatrain_f16.err
head -45 $O/r03_step_breakdown_metatrain_f16.csv
