#!/bin/bash
# round 6, call 41: the whole GPU suite on the final tree (incl. tests/test_e2_full_gpu.py) with the parity exports
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06u; mkdir -p $O; R=r06
LP_PARITY_OUT=$O timeout 1700 python -m pytest tests -m gpu -q -s > $O/${R}_pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee $O/summary.txt
grep -E "passed|failed|error" $O/${R}_pytest_gpu.log | tail -3 | cut -c1-300 | tee $O/${R}_pytest_gpu_tail.txt
grep -E "^\.*\[parity|^\.*\[e1|^\.*\[e2|^\.*\[replicas|^\.*\[streams" $O/${R}_pytest_gpu.log | cut -c1-1500 > $O/${R}_parity_lines.txt
