#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/c12; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_generator_module.py tests/test_train_entry_gpu.py tests/test_checkpoint_fixture.py -m gpu -q 2>&1 | tail -4 | cut -c1-300
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also"
run() { tag=$1; shift; env "$@" timeout 300 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; python - $O/bench_$tag.json $tag <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['ms_per_step'], d['value'], (d.get('drive') or {}).get('value'), ((d.get('drive') or {}).get('batch_8') or {}).get('value'))
except Exception as e: print(sys.argv[2], 'ERR',e)
P
}
run base LP_X=0
run ebwd LP_OVERLAP_EBWD=1
run opt LP_OVERLAP_OPTIMIZER=1
run targets2 LP_OVERLAP_TARGETS=2
run wgrad LP_OVERLAP_WGRAD=1
run base2 LP_X=0
for f in $O/*.err; do tail -1 $f | grep -v amdgpu.ids | cut -c1-300; done
