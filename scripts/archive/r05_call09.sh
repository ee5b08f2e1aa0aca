#!/bin/bash
# round 5, call 9: which synchronisation around the critic's power iteration removes the replica divergence (before / after / none)
O=$GRAFT_REPO_ROOT/gpurun_out/r05i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
i=0
for m in none before after; do
  i=$((i+1))
  DIAG_SYNC=$m LP_OVERLAP=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2967$i scripts/dp_replica_diag.py eager 3 128 > $O/diag_$m.log 2>&1
  echo "== one-stream eager DIAG_SYNC=$m rc=$?" | tee -a $O/summary.txt; grep -E "\[replicas\]" $O/diag_$m.log | cut -c1-260 | tee -a $O/summary.txt
done
