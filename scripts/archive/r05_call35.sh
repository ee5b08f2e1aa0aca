#!/bin/bash
# round 5, call 35: sliced border weight gradient: kernel tests, reflection step time, kernel statistics of that step
O=$GRAFT_REPO_ROOT/gpurun_out/r05ad
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_reflect_border.py tests/test_generator_module.py tests/test_discriminator_criterions.py -q -m gpu 2>&1 | tail -3 | tee $O/tests35.txt
timeout 600 python bench.py --padding reflection --steps 30 --warmup 10 --no-cpu-baseline --no-also --no-drive > $O/bench_reflection.json 2> $O/bench_reflection.err
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r05ad/bench_reflection.json').read().strip().split('\n')[-1])
print('reflection', d['ms_per_step'], d['value'], d['config']['launch_mode'], d['config']['padding'])
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o refl -- python $GRAFT_REPO_ROOT/bench.py --padding reflection --steps 3 --warmup 2 --eager --no-cpu-baseline --no-also --no-drive > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -25 "$f" | cut -c1-200 | tee $O/kernel_stats_top.txt
rm -rf $O/prof
