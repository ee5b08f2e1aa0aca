#!/bin/bash
# round 3, GPU call 17: which branches to overlap, and the input path (H2D prefetch on its own stream) beside the multi-stream graph
O=$GRAFT_REPO_ROOT/gpurun_out/r03c17
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
for e in 1 0; do for c in 0 1; do
  LP_OVERLAP_ENCODERS=$e LP_OVERLAP_CRITERIONS=$c timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('metatrain encoders=$e criterions=$c', j['value'], j['ms_per_step'])"
done; done
LP_OVERLAP_ENCODERS=1 LP_OVERLAP_CRITERIONS=0 timeout 300 python scripts/prefetch_overlap_diag.py metatrain 2>&1 | grep "input path"
LP_OVERLAP_ENCODERS=1 LP_OVERLAP_CRITERIONS=0 LP_COPY_PRIORITY=-1 timeout 300 python scripts/prefetch_overlap_diag.py metatrain 2>&1 | grep "input path"
LP_OVERLAP_ENCODERS=1 LP_OVERLAP_CRITERIONS=0 GPU_MAX_HW_QUEUES=8 timeout 300 python scripts/prefetch_overlap_diag.py metatrain 2>&1 | grep "input path"
LP_OVERLAP_ENCODERS=0 LP_OVERLAP_CRITERIONS=0 timeout 300 python scripts/prefetch_overlap_diag.py metatrain 2>&1 | grep "input path"
LP_OVERLAP_CRITERIONS=1 timeout 300 python scripts/prefetch_overlap_diag.py finetune 2>&1 | grep "input path"
LP_OVERLAP_CRITERIONS=1 LP_COPY_PRIORITY=-1 timeout 300 python scripts/prefetch_overlap_diag.py finetune 2>&1 | grep "input path"
LP_OVERLAP_CRITERIONS=0 timeout 300 python scripts/prefetch_overlap_diag.py finetune 2>&1 | grep "input path"
} 2>&1 | tee $O/r03_stream_overlap.txt
