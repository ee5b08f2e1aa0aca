#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/c21; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_conv_pool.py -m gpu -q -s 2>&1 | grep -E "conv-pool|passed|failed|Error|error|assert" | cut -c1-250 | tail -24
timeout 900 python -m pytest tests/test_discriminator_criterions.py tests/test_train_step.py tests/test_streams_gpu.py -m gpu -q 2>&1 | tail -5 | cut -c1-300
timeout 600 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "discriminator" 2>&1 | grep -E "parity-256|passed|failed" | cut -c1-420
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive"
run() { tag=$1; shift; env "$@" timeout 300 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; python - $O/bench_$tag.json $tag <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['ms_per_step'], d['value'], d['roofline']['frac'])
except Exception as e: print(sys.argv[2], 'ERR',e)
P
}
run cp0 LP_D_CONVPOOL=0
run cp1 LP_D_CONVPOOL=1
run cp0b LP_D_CONVPOOL=0
run cp1b LP_D_CONVPOOL=1
for f in $O/*.err; do tail -1 $f | grep -v amdgpu.ids | cut -c1-300; done
