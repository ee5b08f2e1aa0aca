#!/bin/bash
# round 5, call 8: trace every power iteration of the critic inside the eager one-stream data-parallel steps (W / u checksums before and after)
O=$GRAFT_REPO_ROOT/gpurun_out/r05h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
LP_OVERLAP=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29662 scripts/dp_replica_diag.py eager 3 128 > $O/diag_eager_onestream.log 2>&1
echo "== one-stream eager rc=$?" | tee -a $O/summary.txt; grep -E "\[replicas\]|\[sn-probe\]|\[sn-trace\]" $O/diag_eager_onestream.log | cut -c1-400 | tee -a $O/summary.txt
