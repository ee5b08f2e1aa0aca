#!/bin/bash
# round 5, call 33: ATen call sites of one eager step; a reflection-padding meta-training step at the benchmark's geometry (graphed)
O=$GRAFT_REPO_ROOT/gpurun_out/r05ab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python scripts/aten_sites.py > $O/aten_sites.txt 2> $O/aten_sites.err; tail -3 $O/aten_sites.err; head -150 $O/aten_sites.txt | cut -c1-220
timeout 600 python bench.py --padding reflection --steps 30 --warmup 10 --no-cpu-baseline --no-also --no-drive > $O/bench_reflection.json 2> $O/bench_reflection.err; tail -2 $O/bench_reflection.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r05ab/bench_reflection.json').read().strip().split('\n')[-1])
print('reflection', d['ms_per_step'], d['value'], d['config']['launch_mode'], d['config']['padding'])
PY
