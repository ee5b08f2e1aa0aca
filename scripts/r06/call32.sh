#!/bin/bash
# round 6, call 32: 64 x 128 tiles for the pointwise layers whose 128 x 128 tiling under-fills the chip (LP_CONV1X1_BM64 = workgroup threshold)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06k; mkdir -p $O
for pr in 1 2; do for cfg in "A=1" "LP_CONV1X1_BM64=600" "LP_CONV1X1_BM64=1100" "LP_CONV1X1_BM64=2100"; do
  echo "== PREC=$pr $cfg" >> $O/conv1x1.txt
  env $cfg SHAPES=1x1 PREC=$pr WHAT=conv python scripts/conv_micro.py 2>&1 | grep "conv " >> $O/conv1x1.txt
done; done
cat $O/conv1x1.txt
LP_CONV1X1_BM64=1100 timeout 600 python -m pytest tests/test_hip_ops.py tests/test_resnext_hip.py -x -q -m gpu 2>&1 | tail -2 | tee $O/tests.txt
for i in 1 2; do for cfg in "A=1" "LP_CONV1X1_BM64=1100"; do
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'])" | tee -a $O/ab.txt
done; done
