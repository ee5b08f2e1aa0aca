#!/bin/bash
# round 3, GPU call 33 (last): D passes issued in the reference's order from one fork point -- the tape-based critic parity test, streams, bench
O=$GRAFT_REPO_ROOT/gpurun_out/r03c33
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_full_size_parity.py tests/test_streams_gpu.py tests/test_discriminator_criterions.py -m gpu -q -x > $O/tests.log 2>&1
echo "tests rc=$?" | tee $O/summary.txt
grep -E "passed|failed" $O/tests.log | tail -2
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-200 | head -3
timeout 100 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('metatrain default', j['value'], j['ms_per_step'])"
