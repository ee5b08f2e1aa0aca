#!/bin/bash
# round 5, call 24: victims of running beside the conv kernels; conv_pipe vs conv_dma as the noise
O=$GRAFT_REPO_ROOT/gpurun_out/r05x
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "LP_CONV_PIPE=1" "LP_CONV_PIPE=0" "LP_CONV_PIPE=0 LP_PREC=bf16x3"; do
  env $v timeout 300 python scripts/victim_probe.py 60 2>&1 | grep -v amdgpu.ids | cut -c1-600 | tee -a $O/victims.txt
done
