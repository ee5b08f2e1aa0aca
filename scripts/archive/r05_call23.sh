#!/bin/bash
# round 5, call 23: which kernel of the generator forward disturbs a concurrent power iteration?
O=$GRAFT_REPO_ROOT/gpurun_out/r05w
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in conv16 conv16small actpack instnorm linear gsn; do
  echo "--- noise: $n"
  SN_NOISE=$n timeout 300 python scripts/sn_determinism.py 100 2>&1 | grep -v amdgpu.ids | head -1 | cut -c1-500 | tee $O/sn_beside_$n.txt
done
