#!/bin/bash
# round 5, call 32: reflection padding (border-correction kernels): kernel-level, generator and discriminator fixtures from the reference
O=$GRAFT_REPO_ROOT/gpurun_out/r05ab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_reflect_border.py tests/test_generator_module.py tests/test_discriminator_criterions.py -q -m gpu -s 2>&1 | grep -E "passed|failed|parity|Error|error|assert|FAILED" | cut -c1-700 | tail -40 | tee $O/tests32.txt
