#!/bin/bash
# round 5, call 15: element-level picture of the diverged spectral-norm vectors
O=$GRAFT_REPO_ROOT/gpurun_out/r05o
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
DIAG_SYNC=none LP_OVERLAP=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29691 scripts/dp_replica_diag.py eager 4 128 > $O/diag.log 2>&1
grep -E "\[replicas\]" $O/diag.log | cut -c1-330 | head -60 | tee -a $O/summary.txt
