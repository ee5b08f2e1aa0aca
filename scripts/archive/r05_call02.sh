#!/bin/bash
# round 5, call 2: replica-divergence diagnosis (2 gloo ranks on one GPU), the generator's 16-bit-resident conv outputs (parity + A/B timing),
# the critic's fake->G strict-pass start unit sweep, the three tests fixed after call 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for m in eager graph; do for sp in 1 0; do
  LP_DP_SPLIT=$sp timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2965$sp scripts/dp_replica_diag.py $m 3 128 > $O/diag_${m}_split$sp.log 2>&1
  echo "== diag $m split=$sp rc=$?" | tee -a $O/summary.txt; grep "\[replicas\]" $O/diag_${m}_split$sp.log | cut -c1-600 | tee -a $O/summary.txt
done; done
timeout 900 python -m pytest tests/test_generator_module.py tests/test_fsth_plus.py tests/test_train_entry_gpu.py tests/test_full_size_parity.py tests/test_train_step.py tests/test_hip_ops.py -m gpu -q -s -p no:cacheprovider -k "not discriminator and not vgg" > $O/tests_y16.log 2>&1; echo "y16 tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|^FAILED|^ERROR|parity-256\] prec" $O/tests_y16.log | cut -c1-700 | tee -a $O/summary.txt
LP_G_Y16=0 timeout 600 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -p no:cacheprovider -k "generator" > $O/tests_y16off.log 2>&1
grep -E "passed|failed|parity-256\] prec" $O/tests_y16off.log | cut -c1-700 | tee -a $O/summary.txt
for v in "LP_G_Y16=1" "LP_G_Y16=0"; do
  env $v timeout 300 python bench.py --workload generator --steps 100 --warmup 20 --no-cpu-baseline --no-also --no-drive > $O/bench_gen_$v.json 2> $O/bench_gen_$v.err
  echo "generator $v $(python -c "import json;j=json.load(open('$O/bench_gen_$v.json'));print(j['ms_per_step'], j['value'])" 2>&1)" | tee -a $O/summary.txt
  env $v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-also --no-drive > $O/bench_$v.json 2> $O/bench_$v.err
  echo "metatrain $v $(python -c "import json;j=json.load(open('$O/bench_$v.json'));print(j['ms_per_step'], j['value'], j['roofline']['frac'])" 2>&1)" | tee -a $O/summary.txt
done
for f in 1 2 3; do
  LP_D_GPASS_FROM=$f timeout 600 python tests/test_metatrain_full_gpu.py $O/par_from$f.json > $O/par_from$f.log 2>&1
  python - <<PY | tee -a $O/summary.txt
import json
try:
    r = json.load(open('$O/par_from$f.json'))
    print('FROM=$f', {k: float(f'{v:.3g}') for k, v in r['errors'].items() if k in ('fake_score_G', 'loss.adversarial_G', 'fake_rgbs', 'fake_segm')})
except Exception as e:
    print('FROM=$f failed', e)
PY
  LP_D_GPASS_FROM=$f timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-also --no-drive > $O/bench_from$f.json 2> $O/bench_from$f.err
  echo "FROM=$f $(python -c "import json;j=json.load(open('$O/bench_from$f.json'));print(j['ms_per_step'], j['value'])" 2>&1)" | tee -a $O/summary.txt
done
