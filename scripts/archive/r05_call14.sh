#!/bin/bash
# round 5, call 14: 10-step runs -- HIP power iteration (plain / with stream-ordered snapshots) vs a torch power iteration for the critic's conv layers
O=$GRAFT_REPO_ROOT/gpurun_out/r05n
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
i=20
for v in "DIAG_SYNC=none" "DIAG_SYNC=clone" "DIAG_TORCH_SN=1 DIAG_SYNC=none" "DIAG_SYNC=none LP_OVERLAP_X=1"; do
  i=$((i+1))
  tag=$(echo $v | tr '= ' '__')
  env $v LP_OVERLAP=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 296$i scripts/dp_replica_diag.py eager 10 128 > $O/diag_$tag.log 2>&1
  echo "== $v rc=$? : $(grep -E "\[replicas\] after eager step" $O/diag_$tag.log | sed -E 's/.*step ([0-9]+): ([0-9]+) of.*/s\1:\2/' | tr '\n' ' ')" | tee -a $O/summary.txt
done
