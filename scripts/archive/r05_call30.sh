#!/bin/bash
# round 5, call 30: packed-fp32 forms incl. v_pk_mov / v_fma_mix; the fixed library: concurrency exactness test, replica tests, replica diagnosis
O=$GRAFT_REPO_ROOT/gpurun_out/r05z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python scripts/pk_forms_probe.py 4 2>&1 | grep -v amdgpu.ids | cut -c1-1200 | tee $O/pk_forms.txt
timeout 600 python -m pytest tests/test_concurrent_exactness_gpu.py tests/test_data_parallel_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee $O/tests30.txt
timeout 300 python scripts/victim_probe.py 40 2>&1 | grep -v amdgpu.ids | grep victims | cut -c1-400 | tee $O/victims30.txt
