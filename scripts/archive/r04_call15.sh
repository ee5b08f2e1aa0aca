#!/bin/bash
# round 4, GPU call 15: bf16x3 1x1 conv: registers capped for 3+ waves per SIMD (127 VGPRs instead of 170) x one / two A buffers -- micro + step
O=$GRAFT_REPO_ROOT/gpurun_out/r04c15
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/latent_pose_reenactment_amd/build/ab/liblp_hip_x3w3.so
for v in "A=1" "LP_CONV_ADBUF=0" "LP_LIB_OVERRIDE=$V" "LP_LIB_OVERRIDE=$V LP_CONV_ADBUF=0"; do
  echo "== $v" | sed 's#/tmp/[^ ]*/latent#latent#' >> $O/micro.log
  env $v SHAPES=1x1 PREC=1 WHAT=conv REPS=30 timeout 120 python scripts/conv_micro.py 2>&1 | grep -v amdgpu >> $O/micro.log
done
cat $O/micro.log
for v in "A=1" "LP_LIB_OVERRIDE=$V" "LP_LIB_OVERRIDE=$V LP_CONV_ADBUF=0" "A=1"; do
  env $v timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/b.json 2> $O/b.err
  echo "$v" | sed 's#/tmp/[^ ]*/latent#latent#'; python -c "
import json; j=json.load(open('$O/b.json')); print('   ', j['ms_per_step'], 'ms', j['value'], 'img/s')"
done
