#!/bin/bash
# round 5, call 26: which stage of the power iteration is disturbed beside the conv kernels?
O=$GRAFT_REPO_ROOT/gpurun_out/r05z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LP_SN_DEBUG_SYNC=1 timeout 300 python scripts/victim_probe.py 4 2>&1 | grep -v amdgpu.ids | cut -c1-1200 | tee -a $O/stages.txt
