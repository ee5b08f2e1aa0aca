#!/bin/bash
# round 5, call 25: host synchronisation after every kernel of the power iteration -- does the disturbance beside conv kernels persist?
O=$GRAFT_REPO_ROOT/gpurun_out/r05y
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "LP_SN_DEBUG_SYNC=0" "LP_SN_DEBUG_SYNC=1"; do
  echo "--- $v"
  env $v timeout 300 python scripts/victim_probe.py 60 2>&1 | grep -v amdgpu.ids | cut -c1-600 | tee -a $O/victims.txt
done
