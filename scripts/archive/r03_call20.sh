#!/bin/bash
# round 3, GPU call 20: pinned-slot fill by single-threaded memmove vs Tensor.copy_ (intra-op thread pool), same box
O=$GRAFT_REPO_ROOT/gpurun_out/r03c20
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
nproc; python -c "import torch; print('torch threads', torch.get_num_threads())"
LP_PIN_COPY=torch timeout 300 python scripts/prefetch_overlap_diag.py finetune 2>&1 | grep "input path" | sed 's/^/copy_: /'
LP_PIN_COPY=torch OMP_WAIT_POLICY=passive timeout 300 python scripts/prefetch_overlap_diag.py finetune 2>&1 | grep "input path" | sed 's/^/copy_ + OMP_WAIT_POLICY=passive: /'
LP_PIN_COPY=memmove timeout 300 python scripts/prefetch_overlap_diag.py finetune 2>&1 | grep "input path" | sed 's/^/memmove: /'
LP_PIN_COPY=memmove timeout 300 python scripts/prefetch_overlap_diag.py metatrain 2>&1 | grep "input path" | sed 's/^/memmove: /'
timeout 300 python -m pytest tests/test_prefetch.py -m gpu -q -s 2>&1 | grep -E "input path\]|passed|failed"
} 2>&1 | tee $O/r03_input_path.txt
