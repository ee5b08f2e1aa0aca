#!/bin/bash
# round 3, GPU call 32: streams.join_all after every backward pass: the data-parallel re-cut capture with the D passes on side streams
O=$GRAFT_REPO_ROOT/gpurun_out/r03c32
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_data_parallel_gpu.py tests/test_streams_gpu.py tests/test_train_entry_gpu.py -m gpu -q -s > $O/tests.log 2>&1
echo "tests rc=$?" | tee $O/summary.txt
grep -E "\[streams\]|\[dp\]|passed|failed" $O/tests.log | cut -c1-300 | tail -8
grep -E "^FAILED|^ERROR|Unjoined|unjoined" $O/tests.log | cut -c1-300 | head -5
