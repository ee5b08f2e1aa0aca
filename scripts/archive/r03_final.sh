#!/bin/bash
# round 3, final check of HEAD on the GPU box: the whole GPU test suite, smoke(), the default bench line
O=$GRAFT_REPO_ROOT/gpurun_out/r03f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x > $O/r03_pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?" | tee $O/summary.txt
tail -3 $O/r03_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r03_smoke.log 2>&1
echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 1200 python bench.py > $O/r03_bench_f16.json 2> $O/r03_bench_f16.err
echo "bench rc=$?" | tee -a $O/summary.txt
python -c "
import json; j=json.load(open('$O/r03_bench_f16.json')); print(j['value'], j['ms_per_step'], j['finetune_step']['ms_per_step'], j['strict_mode_bf16x3']['ms_per_step'], j['cpu_baseline'].get('value'), j['drive'].get('value'))"
