#!/bin/bash
# round 5, call 17: is the power iteration still bitwise repeatable when TWO processes share the GPU (the gloo data-parallel tests' setting)?
O=$GRAFT_REPO_ROOT/gpurun_out/r05q
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 300 python scripts/sn_determinism.py 150 2>&1 | grep -v amdgpu.ids | cut -c1-1200 > $O/proc_a.txt) &
(timeout 300 python scripts/sn_determinism.py 150 2>&1 | grep -v amdgpu.ids | cut -c1-1200 > $O/proc_b.txt) &
wait
echo "--- process A"; cat $O/proc_a.txt; echo "--- process B"; cat $O/proc_b.txt
echo "--- alone"; timeout 300 python scripts/sn_determinism.py 150 2>&1 | grep -v amdgpu.ids | cut -c1-600 | tee $O/alone.txt
