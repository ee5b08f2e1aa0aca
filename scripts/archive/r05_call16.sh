#!/bin/bash
# round 5, call 16: does lp_sn_power_iter depend on the previous contents of its buffer sets when W changes between calls?
O=$GRAFT_REPO_ROOT/gpurun_out/r05p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python scripts/sn_determinism.py 20 2>&1 | grep -v amdgpu.ids | cut -c1-1500 | tee $O/sn_determinism.txt
