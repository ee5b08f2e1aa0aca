#!/bin/bash
# round 5, call 12: in which phase of a data-parallel step do the critic's spectral-norm vectors part between ranks?
O=$GRAFT_REPO_ROOT/gpurun_out/r05l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
DIAG_PHASES=1 DIAG_SYNC=none LP_OVERLAP=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29681 scripts/dp_replica_diag.py eager 3 128 > $O/diag.log 2>&1
echo "rc=$?" | tee -a $O/summary.txt; grep -E "\[replicas\]|\[phase\]" $O/diag.log | cut -c1-300 | tee -a $O/summary.txt
