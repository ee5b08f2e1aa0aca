#!/bin/bash
# round 6, call 40: tie-masked full-depth gradient test of the pose encoder (E2)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06t; mkdir -p $O; cp profiles/r06_parity_gradients_*.json $O/
LP_PARITY_OUT=$O timeout 900 python -m pytest tests/test_e2_full_gpu.py -x -q -m gpu -s 2>&1 | grep -E "e2-full|passed|failed|Error|assert" | cut -c1-1500 | tee $O/tests.txt
