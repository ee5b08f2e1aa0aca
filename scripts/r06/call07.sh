#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/c07; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 | cut -c1-400 > $O/pytest.txt
cat $O/pytest.txt
