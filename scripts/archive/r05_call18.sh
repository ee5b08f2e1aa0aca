#!/bin/bash
# round 5, call 18: which rank's power iteration is off its own fp64 evaluation when the replicas part?
O=$GRAFT_REPO_ROOT/gpurun_out/r05s
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
DIAG_SYNC=clone LP_OVERLAP=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29691 scripts/dp_replica_diag.py eager 6 128 > $O/diag.log 2>&1
grep -E "\[sn-clones\]" $O/diag.log | grep -v "differ \[\]; sigma differs \[\]" | cut -c1-520 | head -24 | tee -a $O/summary.txt
grep -E "worst" $O/diag.log | head -8 | cut -c1-200
