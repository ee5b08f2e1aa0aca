#!/bin/bash
# round 5, call 11: probe of the label embedding's power iteration across ranks (no training needed for the first probe)
O=$GRAFT_REPO_ROOT/gpurun_out/r05k
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
DIAG_SYNC=none LP_OVERLAP=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29681 scripts/dp_replica_diag.py eager 2 128 > $O/diag.log 2>&1
echo "rc=$?" | tee -a $O/summary.txt; grep -E "\[replicas\]|\[sn-probe\]|\[embed-probe\]" $O/diag.log | cut -c1-500 | tee -a $O/summary.txt
