#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/c08; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_generator_module.py tests/test_fsth_plus.py tests/test_train_step.py tests/test_full_size_parity.py -m gpu -q -k "not discriminator and not vgg" 2>&1 | tail -8 | cut -c1-400
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive"
run() { tag=$1; shift; env "$@" timeout 300 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; python - $O/bench_$tag.json $tag <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))
except Exception as e: print(sys.argv[2], 'ERR',e)
P
}
run old LP_PLANES_DIRECT=0 LP_G_RAW16=0
run new LP_X=0
run old2 LP_PLANES_DIRECT=0 LP_G_RAW16=0
run new2 LP_X=0
timeout 300 python bench.py --workload generator --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('generator new', d['ms_per_step'])"
LP_PLANES_DIRECT=0 LP_G_RAW16=0 timeout 300 python bench.py --workload generator --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('generator old', d['ms_per_step'])"
for f in $O/*.err; do tail -1 $f | grep -v amdgpu.ids | cut -c1-300; done
