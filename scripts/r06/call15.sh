#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_phase.py -m gpu -q -s -x 2>&1 | grep -E "phase-conv|passed|failed|Error|assert|fault" | cut -c1-260 | tail -40
