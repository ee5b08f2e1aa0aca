#!/bin/bash
# round 5, call 39: second full run of the GPU suite on the final tree (flakiness check) + smoke
O=$GRAFT_REPO_ROOT/gpurun_out/r05ae
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_2.log 2>&1; echo "pytest -m gpu rc=$?" | tee $O/summary.txt
grep -E "passed|failed|error" $O/pytest_gpu_2.log | tail -3 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_2.txt 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
