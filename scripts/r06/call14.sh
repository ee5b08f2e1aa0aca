#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/c14; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in 1 2; do
LP_CONV_X3_BN64=$v PREC=1 WHAT=conv timeout 200 python scripts/conv_micro.py 2>&1 | grep prec > $O/x3_mid_$v.txt
LP_CONV_X3_BN64=$v SHAPES=small PREC=1 WHAT=conv timeout 200 python scripts/conv_micro.py 2>&1 | grep prec >> $O/x3_mid_$v.txt
done
paste -d'|' $O/x3_mid_1.txt $O/x3_mid_2.txt | cut -c1-230
LP_CONV_X3_BN64=2 timeout 600 python -m pytest tests/test_conv_stats.py tests/test_generator_module.py tests/test_discriminator_criterions.py -m gpu -q 2>&1 | tail -3 | cut -c1-300
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive"
run() { tag=$1; shift; env "$@" timeout 300 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; python - $O/bench_$tag.json $tag <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['ms_per_step'], d['value'])
except Exception as e: print(sys.argv[2], 'ERR',e)
P
}
run m1 LP_CONV_X3_BN64=1
run m2 LP_CONV_X3_BN64=2
run m1b LP_CONV_X3_BN64=1
run m2b LP_CONV_X3_BN64=2
run opt LP_OVERLAP_OPTIMIZER=1
run m1c LP_CONV_X3_BN64=1
for f in $O/*.err; do tail -1 $f | grep -v amdgpu.ids | cut -c1-300; done
