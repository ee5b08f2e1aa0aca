#!/bin/bash
# round 3, GPU call 14: 16-bit-resident conv outputs in the embedder's fp16 mode -- per-op and whole-net parity, step A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r03c14
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_resnext_hip.py tests/test_mobilenet_train_hip.py -m gpu -q -s --maxfail=20 > $O/tests_e.log 2>&1
echo "E tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|error" $O/tests_e.log | tail -2
grep -E "^FAILED|^ERROR|\[parity\] (shallow|resnext50)|bn_bwd16_h" $O/tests_e.log | cut -c1-330
timeout 900 python -m pytest tests/test_metatrain_step.py tests/test_conv_stats.py tests/test_abi.py -m gpu -q -s --maxfail=20 > $O/tests_m.log 2>&1
echo "metatrain tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|error" $O/tests_m.log | tail -2
grep -E "^FAILED|^ERROR|\[parity\] meta" $O/tests_m.log | cut -c1-600
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_metatrain.json 2> $O/bench_metatrain.err
LP_E_Y16=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_metatrain_y32.json 2> $O/bench_metatrain_y32.err
python -c "
import json
for f in ('bench_metatrain', 'bench_metatrain_y32'):
    try:
        j=json.load(open('$O/%s.json' % f)); print(f, j['value'], j['ms_per_step'], {k: (v.get('achieved'), v.get('unit')) for k, v in j.items() if k.startswith('roofline_')})
    except Exception as e: print(f, 'ERR', e)"
tail -3 $O/bench_metatrain.err
