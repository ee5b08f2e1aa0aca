#!/bin/bash
# round 5, call 10: bisect the replica divergence of the critic's spectral-norm vectors (one-stream eager data-parallel steps, 2 gloo ranks on one GPU)
O=$GRAFT_REPO_ROOT/gpurun_out/r05j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
i=10
for v in "LP_NONE=1" "LP_SN_DEFER=0" "LP_D_POOL_RELU=0" "LP_D_GPASS_PREC=f16" "LP_FUSED_ACCUM=0" "LP_DP_SPLIT=0" "LP_PREC=bf16x3" "DIAG_NO_REDUCER=1" "HIP_LAUNCH_BLOCKING=1" "AMD_SERIALIZE_KERNEL=3"; do
  i=$((i+1))
  tag=$(echo $v | tr '=' '_')
  env $v DIAG_SYNC=none LP_OVERLAP=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 296$i scripts/dp_replica_diag.py eager 4 128 > $O/diag_$tag.log 2>&1
  echo "== $v rc=$? : $(grep -E "\[replicas\] after eager step" $O/diag_$tag.log | sed -E 's/.*step ([0-9]+): ([0-9]+) of.*/s\1:\2/' | tr '\n' ' ')" | tee -a $O/summary.txt
done
