#!/bin/bash
# round 3, GPU call 10: 64-channel chunks for the 1x1 contractions (A/B + parity), re-run of the re-gated parity tests
O=$GRAFT_REPO_ROOT/gpurun_out/r03c10
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cc in 32 64; do
  LP_CONV_CC1=$cc SHAPES=1x1 PREC=2 WHAT=conv REPS=30 timeout 120 python scripts/conv_micro.py > $O/conv1x1_f16_cc$cc.txt 2>&1
  LP_CONV_CC1=$cc SHAPES=1x1 PREC=1 WHAT=conv REPS=30 timeout 120 python scripts/conv_micro.py > $O/conv1x1_x3_cc$cc.txt 2>&1
done
paste -d'|' $O/conv1x1_f16_cc32.txt $O/conv1x1_f16_cc64.txt $O/conv1x1_x3_cc32.txt $O/conv1x1_x3_cc64.txt | awk -F'|' '{print $1 "|" $2 "|" $4 "|" $6 "|" $8}' > $O/r03_conv1x1_cc64.txt
cat $O/r03_conv1x1_cc64.txt | cut -c1-220
timeout 1500 python -m pytest tests/test_hip_ops.py tests/test_conv_stats.py tests/test_resnext_hip.py tests/test_mobilenet_train_hip.py tests/test_train_step.py tests/test_metatrain_step.py tests/test_full_size_parity.py tests/test_data_parallel_gpu.py -m gpu -q -s --maxfail=80 > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt
for cc in 32 64; do
  LP_CONV_CC1=$cc timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_metatrain_cc$cc.json 2> $O/bench_metatrain_cc$cc.err
done
grep -E "passed|failed|error" $O/tests.log | tail -3
grep -E "FAILED|\[parity\] (meta-train step|train step)|\[parity-256\]|\[dp\]" $O/tests.log | cut -c1-600
python -c "
import json
for cc in (32, 64):
    j=json.load(open('$O/bench_metatrain_cc%d.json' % cc)); print(cc, j['value'], j['ms_per_step'], {k: (v.get('achieved'), v.get('unit'), v.get('frac')) for k, v in j.items() if k.startswith('roofline')})"
