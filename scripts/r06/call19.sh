#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_train_step.py tests/test_fsth_plus.py tests/test_checkpoint_fixture.py -m gpu -q -x 2>&1 | grep -E "Error|error|assert|FAILED|lp_" | head -20 | cut -c1-400
