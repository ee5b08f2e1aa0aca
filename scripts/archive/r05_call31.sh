#!/bin/bash
# round 5, call 31: after the packed-fp32 fix: victims, default bench line, configs[2] parity (default + bf16x3), ATen inventory
O=$GRAFT_REPO_ROOT/gpurun_out/r05aa
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python scripts/victim_probe.py 40 2>&1 | grep -v amdgpu.ids | grep -E "victims|stages" | cut -c1-300 | tee $O/victims.txt
timeout 600 python bench.py --steps 100 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r05aa/bench_default.json').read().strip().split('\n')[-1])
print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('kernels_per_step'))
PY
LP_PARITY_OUT=$O timeout 900 python -m pytest tests/test_metatrain_full_gpu.py -x -q -m gpu -s 2>&1 | grep -E "passed|failed|rel|parity" | cut -c1-900 | tail -12 | tee $O/parity.txt
timeout 300 python scripts/aten_ops.py > $O/aten_ops.txt 2>&1; head -45 $O/aten_ops.txt | cut -c1-600
