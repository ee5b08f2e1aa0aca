#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/c09; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in 0 1; do
LP_CONV_X3_BN64=$v SHAPES=small PREC=1 WHAT=conv timeout 200 python scripts/conv_micro.py > $O/small_x3_bn64_$v.txt 2>&1
done
paste -d'|' $O/small_x3_bn64_0.txt $O/small_x3_bn64_1.txt | cut -c1-240
LP_CONV_X3_BN64=1 timeout 600 python -m pytest tests/test_conv_stats.py tests/test_hip_ops.py tests/test_generator_module.py -m gpu -q 2>&1 | tail -4 | cut -c1-300
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive"
run() { tag=$1; shift; env "$@" timeout 300 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; python - $O/bench_$tag.json $tag <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))
except Exception as e: print(sys.argv[2], 'ERR',e)
P
}
run bn64_0 LP_CONV_X3_BN64=0
run bn64_1 LP_CONV_X3_BN64=1
run bn64_0b LP_CONV_X3_BN64=0
run bn64_1b LP_CONV_X3_BN64=1
for f in $O/*.err; do tail -1 $f | grep -v amdgpu.ids | cut -c1-300; done
