#!/bin/bash
# round 3, GPU call 19: input-path regression hunt: the same prefetch test in the tree of commit 7ed75e5 (passed on call 10) and here
O=$GRAFT_REPO_ROOT/gpurun_out/r03c19
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
{
cd $GRAFT_REPO_ROOT/_old && timeout 300 python -m pytest tests/test_prefetch.py -m gpu -q -s 2>&1 | grep -E "input path\]|passed|failed" | sed 's/^/OLD TREE: /'
cd $GRAFT_REPO_ROOT && timeout 300 python -m pytest tests/test_prefetch.py -m gpu -q -s 2>&1 | grep -E "input path\]|passed|failed" | sed 's/^/NEW TREE: /'
cd $GRAFT_REPO_ROOT && timeout 400 python scripts/prefetch_diag.py 2>&1 | grep -v amdgpu | tail -14
} 2>&1 | tee $O/r03_input_path_bisect.txt
