#!/bin/bash
# round 5, call 20: power-iteration repeatability with two BUSY processes time-slicing the GPU (long runs, noise kernels on a side stream)
O=$GRAFT_REPO_ROOT/gpurun_out/r05t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 400 python scripts/sn_determinism.py 1500 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > $O/proc_a.txt) &
(timeout 400 python scripts/sn_determinism.py 1500 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > $O/proc_b.txt) &
wait
echo "--- process A"; cat $O/proc_a.txt; echo "--- process B"; cat $O/proc_b.txt
