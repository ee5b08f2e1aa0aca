#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/c18; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_generator_module.py tests/test_fsth_plus.py tests/test_train_step.py tests/test_hip_ops.py tests/test_checkpoint_fixture.py tests/test_train_entry_gpu.py -m gpu -q 2>&1 | tail -5 | cut -c1-300
timeout 600 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "generator" 2>&1 | grep -E "parity-256|passed|failed" | cut -c1-330
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive"
run() { tag=$1; shift; env "$@" timeout 300 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; python - $O/bench_$tag.json $tag <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['ms_per_step'], d['value'], d['roofline']['frac'])
except Exception as e: print(sys.argv[2], 'ERR',e)
P
}
run ph0 LP_G_PHASE=0
run ph1 LP_G_PHASE=1
run ph0b LP_G_PHASE=0
run ph1b LP_G_PHASE=1
for v in 0 1; do LP_G_PHASE=$v timeout 300 python bench.py --workload generator --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('generator LP_G_PHASE=$v', d['ms_per_step'])"; done
for f in $O/*.err; do tail -1 $f | grep -v amdgpu.ids | cut -c1-300; done
