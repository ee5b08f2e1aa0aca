#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/c10; mkdir -p $O
cd $GRAFT_REPO_ROOT
LP_WGRAD3_X3=1 timeout 900 python -m pytest tests/test_wgrad3_pipe.py -m gpu -q -s 2>&1 | grep -E "dense bf16x3|passed|failed|FAILED" | cut -c1-200 | tail -40
for v in 0 1; do
LP_WGRAD3_X3=$v SHAPES=wgrad PREC=1 WHAT=wgrad timeout 300 python scripts/conv_micro.py > $O/wgrad_x3_$v.txt 2>&1
done
paste -d'|' $O/wgrad_x3_0.txt $O/wgrad_x3_1.txt | cut -c1-230
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive"
run() { tag=$1; shift; env "$@" timeout 300 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; python - $O/bench_$tag.json $tag <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))
except Exception as e: print(sys.argv[2], 'ERR',e)
P
}
run w0 LP_WGRAD3_X3=0
run w1 LP_WGRAD3_X3=1
run w0b LP_WGRAD3_X3=0
run w1b LP_WGRAD3_X3=1
for f in $O/*.err; do tail -1 $f | grep -v amdgpu.ids | cut -c1-300; done
