#!/bin/bash
# round 4, GPU call 12: conv_pipe after the bf16x3 fix (builtin MFMAs there): parity; micro A/B default vs s_setprio(1) around the MFMA groups
O=$GRAFT_REPO_ROOT/gpurun_out/r04c12
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_conv_pipe.py -m gpu -q -x -s > $O/pipe_tests.log 2>&1; echo "pipe tests rc=$?" | tee $O/summary.txt
grep -E "passed|failed|FAILED" $O/pipe_tests.log | tail -5
for v in "A=1" "LP_LIB_OVERRIDE=$GRAFT_REPO_ROOT/latent_pose_reenactment_amd/build/ab/liblp_hip_setprio.so" "A=1" "LP_LIB_OVERRIDE=$GRAFT_REPO_ROOT/latent_pose_reenactment_amd/build/ab/liblp_hip_setprio.so"; do
  echo "== $v" | sed 's#/tmp/[^ ]*/latent#latent#' >> $O/micro.log
  env $v PREC=2 WHAT=conv REPS=30 timeout 120 python scripts/conv_micro.py 2>&1 | grep -E "32, 32|64, 64, 256, 256|128, 128, 128|256, 256, 64, 64|256, 256, 64, 128" >> $O/micro.log
done
echo "== PREC=1" >> $O/micro.log; PREC=1 WHAT=conv REPS=20 timeout 120 python scripts/conv_micro.py 2>&1 | grep -E "32, 32|64, 64, 256, 256|128, 128, 128" >> $O/micro.log
cat $O/micro.log
