#!/bin/bash
# round 3, GPU call 12: dedicated 1x1 weight-gradient kernel -- parity, A/B against the generic kernel, step bench + per-shape table
O=$GRAFT_REPO_ROOT/gpurun_out/r03c12
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_resnext_hip.py tests/test_mobilenet_train_hip.py -m gpu -q -x -k "pointwise or flat_1x1 or mobilenet" > $O/tests_first.log 2>&1
echo "first tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/tests_first.log
for v in new old; do
  env=""; [ $v = old ] && env="LP_WGRAD1X1_OLD=1"
  env $env SHAPES=1x1 PREC=2 WHAT=wgrad REPS=30 timeout 120 python scripts/conv_micro.py > $O/wgrad1x1_f16_$v.txt 2>&1
  env $env SHAPES=1x1 PREC=1 WHAT=wgrad REPS=30 timeout 120 python scripts/conv_micro.py > $O/wgrad1x1_x3_$v.txt 2>&1
done
paste -d'|' $O/wgrad1x1_f16_old.txt $O/wgrad1x1_f16_new.txt $O/wgrad1x1_x3_old.txt $O/wgrad1x1_x3_new.txt | awk -F'|' '{print $1 "|" $2 "|" $4 "|" $6 "|" $8}' | grep -v amdgpu > $O/r03_wgrad1x1.txt
cat $O/r03_wgrad1x1.txt | cut -c1-220
timeout 1500 python -m pytest tests/test_hip_ops.py tests/test_resnext_hip.py tests/test_mobilenet_train_hip.py tests/test_train_step.py tests/test_metatrain_step.py tests/test_generator_module.py -m gpu -q --maxfail=80 > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive --shapes $O/shapes_metatrain.csv > $O/bench_metatrain.json 2> $O/bench_metatrain.err
LP_WGRAD1X1_OLD=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_metatrain_old.json 2> $O/bench_metatrain_old.err
grep -E "passed|failed|error" $O/tests.log | tail -3
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-300
python -c "
import json
for f in ('bench_metatrain', 'bench_metatrain_old'):
    try:
        j=json.load(open('$O/%s.json' % f)); print(f, j['value'], j['ms_per_step'], {k: (v.get('achieved'), v.get('unit')) for k, v in j.items() if k.startswith('roofline_')})
    except Exception as e: print(f, 'ERR', e)"
