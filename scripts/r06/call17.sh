#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_phase.py -m gpu -q -s 2>&1 | grep -E "phase-dgrad|passed|failed|Error|assert|fault" | cut -c1-220 | tail -30
for pr in 1 2; do PREC=$pr timeout 200 python scripts/r06/phase_micro.py 2>&1 | grep prec; done
