#!/bin/bash
# round 5, call 1: the refactored tree (one backend, DP buckets, plain-error parity) on the GPU: full suite + parity exports, the critic's
# fake->G pass precision experiment (plain error of fake_score_G / adversarial_G and its cost), ATen inventory of a meta-training step
O=$GRAFT_REPO_ROOT/gpurun_out/r05a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LP_PARITY_OUT=$O timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/tests.log 2>&1; echo "gpu tests rc=$?" | tee $O/summary.txt
grep -E "passed|failed|error" $O/tests.log | tail -3 | tee -a $O/summary.txt
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-300 | head -30 | tee -a $O/summary.txt
grep -E "parity-configs2|e1-full|few|frame\(s\)" $O/tests.log | cut -c1-1200 | tee -a $O/summary.txt
for v in "LP_D_GPASS_PREC=f16" "LP_D_GPASS_FROM=4" "LP_D_GPASS_FROM=6"; do
  tag=$(echo $v | tr '=' '_')
  env $v timeout 600 python tests/test_metatrain_full_gpu.py $O/par_$tag.json > $O/par_$tag.log 2>&1
  python - <<PY | tee -a $O/summary.txt
import json
try:
    r = json.load(open('$O/par_$tag.json'))
    print('$v', {k: float(f'{v:.3g}') for k, v in r['errors'].items() if k in ('fake_score_G', 'loss.adversarial_G', 'fake_rgbs', 'embeds', 'loss.feature_matching')}, r['conditioned'])
except Exception as e:
    print('$v failed', e)
PY
done
for v in "LP_D_GPASS_PREC=bf16x3" "LP_D_GPASS_PREC=f16" "LP_D_GPASS_FROM=4"; do
  tag=$(echo $v | tr '=' '_')
  env $v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-also --no-drive > $O/bench_$tag.json 2> $O/bench_$tag.err
  echo "$v $(python -c "import json;j=json.load(open('$O/bench_$tag.json'));print(j['ms_per_step'], j['value'])" 2>&1)" | tee -a $O/summary.txt
done
timeout 300 python scripts/aten_ops.py > $O/aten_ops_metatrain.txt 2> $O/aten_ops.err; echo "aten rc=$?" | tee -a $O/summary.txt
head -45 $O/aten_ops_metatrain.txt | cut -c1-400
