#!/bin/bash
# round 5, call 7: where do the replicas' spectral-norm vectors part?  (probe: one extra power iteration from compared inputs)
O=$GRAFT_REPO_ROOT/gpurun_out/r05g
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
LP_OVERLAP=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29662 scripts/dp_replica_diag.py eager 3 128 > $O/diag_eager_onestream.log 2>&1
echo "== one-stream eager rc=$?" | tee -a $O/summary.txt; grep -E "\[replicas\]|\[sn-probe\]" $O/diag_eager_onestream.log | cut -c1-700 | tee -a $O/summary.txt
DIAG_EAGER_AFTER=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29663 scripts/dp_replica_diag.py graph 3 128 > $O/diag_graph.log 2>&1
echo "== graph rc=$?" | tee -a $O/summary.txt; grep -E "\[replicas\]|\[sn-probe\]" $O/diag_graph.log | cut -c1-700 | tee -a $O/summary.txt
