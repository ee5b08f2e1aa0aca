#!/bin/bash
# round 3, GPU call 13: conv epilogue transposed 16 rows at a time (LDS scratch 69 KB -> 17 KB per workgroup): A/B on the 1x1 and 3x3 layer
# classes, conv parity tests, step bench
O=$GRAFT_REPO_ROOT/gpurun_out/r03c13
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/latent_pose_reenactment_amd/liblp_hip_epiold.so
for v in new old; do
  lib=""; [ $v = old ] && lib=$OLD
  LP_LIB_OVERRIDE=$lib SHAPES=1x1 PREC=2 WHAT=conv REPS=30 timeout 120 python scripts/conv_micro.py > $O/conv1x1_f16_$v.txt 2>&1
  LP_LIB_OVERRIDE=$lib PREC=2 WHAT=conv REPS=30 timeout 120 python scripts/conv_micro.py > $O/conv_f16_$v.txt 2>&1
  LP_LIB_OVERRIDE=$lib SHAPES=1x1 PREC=1 WHAT=conv REPS=30 timeout 120 python scripts/conv_micro.py > $O/conv1x1_x3_$v.txt 2>&1
done
paste -d'|' $O/conv1x1_f16_old.txt $O/conv1x1_f16_new.txt $O/conv1x1_x3_old.txt $O/conv1x1_x3_new.txt | awk -F'|' '{print $1 "|" $2 "|" $4 "|" $6 "|" $8}' | grep -v amdgpu > $O/r03_conv_epilogue_lds.txt
paste -d'|' $O/conv_f16_old.txt $O/conv_f16_new.txt | awk -F'|' '{print $1 "|" $2 "|" $4}' | grep -v amdgpu >> $O/r03_conv_epilogue_lds.txt
cat $O/r03_conv_epilogue_lds.txt | cut -c1-200
timeout 1500 python -m pytest tests/test_hip_ops.py tests/test_conv_stats.py tests/test_kernel_variants.py tests/test_resnext_hip.py tests/test_mobilenet_train_hip.py tests/test_generator_module.py tests/test_full_size_parity.py tests/test_discriminator_criterions.py -m gpu -q --maxfail=80 > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_metatrain.json 2> $O/bench_metatrain.err
LP_LIB_OVERRIDE=$OLD timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_metatrain_old.json 2> $O/bench_metatrain_old.err
timeout 300 python bench.py --workload finetune_step --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_finetune.json 2> $O/bench_finetune.err
LP_LIB_OVERRIDE=$OLD timeout 300 python bench.py --workload finetune_step --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_finetune_old.json 2> $O/bench_finetune_old.err
grep -E "passed|failed|error" $O/tests.log | tail -3
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-300
python -c "
import json
for f in ('bench_metatrain', 'bench_metatrain_old', 'bench_finetune', 'bench_finetune_old'):
    try:
        j=json.load(open('$O/%s.json' % f)); print(f, j['value'], j['ms_per_step'], {k: (v.get('achieved'), v.get('unit')) for k, v in j.items() if k.startswith('roofline')})
    except Exception as e: print(f, 'ERR', e)"
