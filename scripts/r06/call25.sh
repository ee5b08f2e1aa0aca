#!/bin/bash
# round 6, call 25: bf16x3 / f16 pointwise layers of the identity encoder -- A/B of the activation-double-buffer + one-weight-stage schedule
# (LP_CONV1X1_B1 = 1: short contractions, 2: all bf16x3) and of 128 x 64 tiles (LP_CONV1X1_BN64), with a correctness run of the 1x1 parity tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06d; mkdir -p $O
for cfg in "A=1" "LP_CONV1X1_B1=1" "LP_CONV1X1_B1=2" "LP_CONV1X1_BN64=1" "A=2"; do
  echo "== $cfg" >> $O/conv1x1.txt
  env $cfg SHAPES=1x1 PREC=1 WHAT=conv python scripts/conv_micro.py 2>&1 | grep "conv " >> $O/conv1x1.txt
done
for cfg in "LP_CONV1X1_B1=1" "LP_CONV1X1_B1=2" "LP_CONV1X1_BN64=1"; do
  echo "== $cfg" >> $O/tests.txt
  env $cfg timeout 900 python -m pytest tests/test_hip_ops.py tests/test_resnext_hip.py tests/test_mobilenet_train_hip.py -x -q -m gpu 2>&1 | tail -3 >> $O/tests.txt
done
cat $O/conv1x1.txt; cat $O/tests.txt
