#!/bin/bash
# round 4, GPU call 4: 64-channel tiles of the pipelined conv kernel (tests + micro A/B), deeper split-K on the tiny maps, the reference goldens again
O=$GRAFT_REPO_ROOT/gpurun_out/r04c04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_conv_pipe.py -m gpu -q -x -s > $O/pipe_tests.log 2>&1; echo "pipe tests rc=$?" | tee $O/summary.txt
grep -E "passed|failed|FAILED" $O/pipe_tests.log | tail -5
for v in "LP_CONV_PIPE_N64=0" "LP_CONV_PIPE_N64=1"; do
  echo "== $v" >> $O/micro.log
  env $v PREC=2 WHAT=conv REPS=30 timeout 120 python scripts/conv_micro.py 2>&1 | grep -E "256, 256|128, 128, 128" >> $O/micro.log
done
for v in "LP_CONV_KSPLIT=8" "LP_CONV_KSPLIT=16 LP_CONV_SPLIT_WGS=512" "LP_CONV_KSPLIT=32 LP_CONV_SPLIT_WGS=512" "LP_CONV_KSPLIT=32 LP_CONV_SPLIT_WGS=1024"; do
  echo "== $v" >> $O/micro.log
  env $v SHAPES=small PREC=2 WHAT=conv REPS=30 timeout 120 python scripts/conv_micro.py 2>&1 | grep -v amdgpu >> $O/micro.log
done
cat $O/micro.log
timeout 300 python -m pytest tests/test_metatrain_step.py -m gpu -q -s -k "128" > $O/golden128.log 2>&1; echo "golden128 rc=$?" | tee -a $O/summary.txt
grep -E "parity\]|passed|failed|Error" $O/golden128.log | cut -c1-1200 | tail -6
LP_PARITY_OUT=$O timeout 600 python -m pytest tests/test_metatrain_full_gpu.py -m gpu -q -s > $O/full.log 2>&1; echo "full configs2 rc=$?" | tee -a $O/summary.txt
grep -E "parity-configs2|passed|failed|Error" $O/full.log | cut -c1-1500 | tail -6
