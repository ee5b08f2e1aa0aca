#!/bin/bash
# round 3, GPU call 34: the fine-tuning-side tests on the final code (train step goldens, input path, checkpoint + drive)
O=$GRAFT_REPO_ROOT/gpurun_out/r03c34
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 95 python -m pytest tests/test_prefetch.py tests/test_checkpoint_fixture.py tests/test_train_step.py -m gpu -q -x > $O/tests.log 2>&1
echo "tests rc=$?" | tee $O/summary.txt
grep -E "passed|failed" $O/tests.log | tail -2
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-200 | head -3
