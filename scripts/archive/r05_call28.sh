#!/bin/bash
# round 5, call 28: which property of sn_wtu makes it sensitive to the conv kernels running beside it?
O=$GRAFT_REPO_ROOT/gpurun_out/r05z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 3; do
  echo "== LP_SN_WTU_VARIANT=$v" | tee -a $O/variants.txt
  LP_SN_WTU_VARIANT=$v LP_SN_DEBUG_SYNC=1 timeout 300 python scripts/victim_probe.py 2 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tee -a $O/variants.txt
done
