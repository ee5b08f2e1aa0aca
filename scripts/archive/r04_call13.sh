#!/bin/bash
# round 4, GPU call 13: the whole GPU suite + smoke on the current tree
O=$GRAFT_REPO_ROOT/gpurun_out/r04c13
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee $O/summary.txt
tail -25 $O/pytest_gpu.log | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
tail -3 $O/smoke.log
