#!/bin/bash
# round 5, call 29: every packed-fp32 operand-selection form beside matrix-core kernels
O=$GRAFT_REPO_ROOT/gpurun_out/r05z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python scripts/pk_forms_probe.py 4 2>&1 | grep -v amdgpu.ids | cut -c1-900 | tee -a $O/pk_forms.txt
