#!/bin/bash
# round 4, GPU call 3: 128-px reference golden through the HIP encoders, the full configs[2] forward parity test, step breakdown of the new default
O=$GRAFT_REPO_ROOT/gpurun_out/r04c03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_metatrain_step.py -m gpu -q -s -k "128" > $O/golden128.log 2>&1; echo "golden128 rc=$?" | tee $O/summary.txt
grep -E "parity\]|passed|failed|Error|assert" $O/golden128.log | cut -c1-900 | tail -8
LP_PARITY_OUT=$O timeout 600 python -m pytest tests/test_metatrain_full_gpu.py -m gpu -q -s > $O/full.log 2>&1; echo "full configs2 rc=$?" | tee -a $O/summary.txt
grep -E "parity-configs2|passed|failed|Error" $O/full.log | cut -c1-1200 | tail -6
tail -5 $O/full.log | cut -c1-600
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_meta -o r04 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-drive > $O/prof_meta.log 2>&1
python scripts/step_breakdown.py $O/prof_meta/r04_kernel_trace.csv > $O/r04_step_breakdown_metatrain_default_a.csv 2>> $O/prof_meta.log
rm -f $O/prof_meta/r04_kernel_trace.csv
head -45 $O/r04_step_breakdown_metatrain_default_a.csv
