#!/bin/bash
# round 5, call 21: ranks take turns for the critic's power iteration (the other process idle meanwhile) -- 10 steps, twice
O=$GRAFT_REPO_ROOT/gpurun_out/r05u
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2; do
DIAG_SYNC=exclusive LP_OVERLAP=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2969$i scripts/dp_replica_diag.py eager 10 128 > $O/diag_$i.log 2>&1
echo "== exclusive run $i rc=$? : $(grep -E "\[replicas\] after eager step" $O/diag_$i.log | sed -E 's/.*step ([0-9]+): ([0-9]+) of.*/s\1:\2/' | tr '\n' ' ')" | tee -a $O/summary.txt
grep -E "\[replicas\] after eager step" $O/diag_$i.log | tail -1 | cut -c1-400
done
