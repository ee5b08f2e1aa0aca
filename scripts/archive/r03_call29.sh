#!/bin/bash
# round 3, GPU call 29: bit-identity of the multi-stream step against the one-stream step
O=$GRAFT_REPO_ROOT/gpurun_out/r03c29
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_streams_gpu.py -m gpu -q -s > $O/tests.log 2>&1
echo "rc=$?"
grep -E "\[streams\]|passed|failed|Error|assert" $O/tests.log | cut -c1-1500 | tail -12
