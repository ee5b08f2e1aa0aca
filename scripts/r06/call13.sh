#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/c13; mkdir -p $O
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive"
run() { tag=$1; shift; env "$@" timeout 300 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; python - $O/bench_$tag.json $tag <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['ms_per_step'], d['value'])
except Exception as e: print(sys.argv[2], 'ERR',e)
P
}
run e0 LP_OVERLAP_EBWD=0
run e1 LP_OVERLAP_EBWD=1
run e0b LP_OVERLAP_EBWD=0
run e1b LP_OVERLAP_EBWD=1
run e0c LP_OVERLAP_EBWD=0
run e1c LP_OVERLAP_EBWD=1
LP_OVERLAP_EBWD=1 timeout 900 python -m pytest tests/test_streams_gpu.py tests/test_train_entry_gpu.py tests/test_metatrain_step.py -m gpu -q 2>&1 | tail -4 | cut -c1-300
for f in $O/*.err; do tail -1 $f | grep -v amdgpu.ids | cut -c1-300; done
