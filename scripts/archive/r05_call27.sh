#!/bin/bash
# round 5, call 27: canaries beside the conv kernels
O=$GRAFT_REPO_ROOT/gpurun_out/r05z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python scripts/canary_probe.py 6 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee -a $O/canary.txt

