#!/bin/bash
# round 4, GPU call 28: full GPU suite on the current tree
O=$GRAFT_REPO_ROOT/gpurun_out/r04c28
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $O/r04_pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee $O/summary.txt
tail -15 $O/r04_pytest_gpu.log | cut -c1-300
