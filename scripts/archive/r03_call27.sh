#!/bin/bash
# round 3, GPU call 27: 256 x 128 output tiles for the big 1x1 layers (8 waves): micro A/B, conv parity, step A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r03c27
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 128 256; do
  LP_CONV1X1_TILE=$t SHAPES=1x1 PREC=2 WHAT=conv REPS=30 timeout 120 python scripts/conv_micro.py > $O/conv1x1_f16_t$t.txt 2>&1
  LP_CONV1X1_TILE=$t SHAPES=1x1 PREC=1 WHAT=conv REPS=30 timeout 120 python scripts/conv_micro.py > $O/conv1x1_x3_t$t.txt 2>&1
done
paste -d'|' $O/conv1x1_f16_t128.txt $O/conv1x1_f16_t256.txt $O/conv1x1_x3_t128.txt $O/conv1x1_x3_t256.txt | awk -F'|' '{print $1 "|" $2 "|" $4 "|" $6 "|" $8}' | grep -v amdgpu > $O/r03_conv1x1_tile256.txt
cat $O/r03_conv1x1_tile256.txt | cut -c1-200
LP_CONV1X1_TILE=256 timeout 900 python -m pytest tests/test_resnext_hip.py tests/test_conv_stats.py tests/test_hip_ops.py -m gpu -q --maxfail=20 > $O/tests.log 2>&1
echo "tests (tile 256) rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|error" $O/tests.log | tail -2
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-300
for t in 256 128 256 128; do
LP_CONV1X1_TILE=$t timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('metatrain 1x1 tile $t', j['value'], j['ms_per_step'], j['roofline_conv1x1']['achieved'])"
done
