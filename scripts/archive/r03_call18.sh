#!/bin/bash
# round 3, GPU call 18: why does the fine-tuning step lose 19 % with a host batch per step?  (hardware-queue sharing between copy and compute streams?)
O=$GRAFT_REPO_ROOT/gpurun_out/r03c18
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
for sk in 0 1 2 3; do
  SKIP_STREAMS=$sk timeout 300 python scripts/prefetch_overlap_diag.py finetune 2>&1 | grep "input path"
done
GPU_MAX_HW_QUEUES=8 timeout 300 python scripts/prefetch_overlap_diag.py finetune 2>&1 | grep "input path"
GPU_MAX_HW_QUEUES=2 timeout 300 python scripts/prefetch_overlap_diag.py finetune 2>&1 | grep "input path"
timeout 600 python -m pytest tests/test_prefetch.py -m gpu -q -s 2>&1 | grep -E "input path|passed|failed"
} 2>&1 | tee $O/r03_input_path_queues.txt
