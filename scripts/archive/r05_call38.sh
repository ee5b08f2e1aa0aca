#!/bin/bash
# round 5, call 38: per-shape times of the border kernels; their tests
O=$GRAFT_REPO_ROOT/gpurun_out/r05ad
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_reflect_border.py -q -m gpu -x 2>&1 | tail -3 | cut -c1-300
timeout 300 python scripts/reflect_micro.py 2>&1 | grep -v amdgpu.ids | tee $O/reflect_micro.txt
