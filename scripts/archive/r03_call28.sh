#!/bin/bash
# round 3, GPU call 28: 1x1 layers with the 3-deep weight ring forced (B two stages ahead, A one): does prefetch depth matter?
O=$GRAFT_REPO_ROOT/gpurun_out/r03c28
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "nbuf2_cc64:LP_CONV_NBUF=2" "nbuf3_cc64:LP_CONV_NBUF=3" "nbuf3_cc32:LP_CONV_NBUF=3 LP_CONV_CC1=32" "nbuf2_cc32:LP_CONV_NBUF=2 LP_CONV_CC1=32"; do
  tag=${v%%:*}; envs=${v#*:}
  env $envs SHAPES=1x1 PREC=2 WHAT=conv REPS=30 timeout 120 python scripts/conv_micro.py 2>&1 | grep -v amdgpu > $O/conv1x1_$tag.txt
done
paste -d'|' $O/conv1x1_nbuf2_cc64.txt $O/conv1x1_nbuf3_cc64.txt $O/conv1x1_nbuf2_cc32.txt $O/conv1x1_nbuf3_cc32.txt | awk -F'|' '{print $1 "|" $2 "|" $4 "|" $6 "|" $8}' > $O/r03_conv1x1_ring.txt
cat $O/r03_conv1x1_ring.txt | cut -c1-200
