#!/bin/bash
# round 5, call 6: bitwise repeatability of the spectral-norm power iteration; generator parity with the 16-bit-resident outputs off / on
O=$GRAFT_REPO_ROOT/gpurun_out/r05f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python scripts/sn_determinism.py 30 2>&1 | grep -v amdgpu.ids | tee $O/sn_determinism.txt
timeout 600 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -p no:cacheprovider -k "generator" > $O/tests.log 2>&1; echo "gen tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|parity-256\]" $O/tests.log | cut -c1-600 | tee -a $O/summary.txt
