#!/bin/bash
# round 4, GPU call 23: target-image VGG taps as 16-bit planes (LP_VGG_TAPS16), real-pass overlap default, EBWD parity test: tests + step A/B + parity JSONs
O=$GRAFT_REPO_ROOT/gpurun_out/r04c23
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_discriminator_criterions.py tests/test_train_step.py tests/test_streams_gpu.py tests/test_full_size_parity.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt
grep -E "passed|failed|FAILED|Error" $O/tests.log | cut -c1-300 | tail -10
LP_PARITY_OUT=$O timeout 600 python -m pytest tests/test_metatrain_full_gpu.py -m gpu -q -s > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt
grep -E "parity-configs2|passed|failed" $O/parity.log | cut -c1-900 | tail -4
for v in "LP_VGG_TAPS16=0" "LP_VGG_TAPS16=1" "LP_VGG_TAPS16=0" "LP_VGG_TAPS16=1"; do
  env $v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-also --no-drive > $O/b.json 2> $O/b.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open('$O/b.json')); print('$v', d['ms_per_step'], 'ms', d['value'], 'img/s')
except Exception as e:
    print('$v bench failed', e, open('$O/b.err').read()[-2500:])
PY
done
env LP_VGG_TAPS16=1 timeout 300 python bench.py --workload finetune_step --steps 40 --warmup 10 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('finetune taps16=1', d['ms_per_step'])" | tee -a $O/summary.txt
env LP_VGG_TAPS16=0 timeout 300 python bench.py --workload finetune_step --steps 40 --warmup 10 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('finetune taps16=0', d['ms_per_step'])" | tee -a $O/summary.txt
