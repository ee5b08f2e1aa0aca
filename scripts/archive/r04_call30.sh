#!/bin/bash
# round 4, GPU call 30: planes-only generated-image VGG pass (LP_VGG_FAKE16): unit test, criterion / step / parity tests, A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r04c30
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_vgg_planes.py tests/test_hip_ops.py tests/test_discriminator_criterions.py tests/test_train_step.py tests/test_full_size_parity.py tests/test_metatrain_full_gpu.py -m gpu -q -s > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt
grep -E "vgg-planes|parity-configs2\] mode default|passed|failed|FAILED|Error" $O/tests.log | cut -c1-420 | tail -12
for v in 0 1 0 1; do LP_VGG_FAKE16=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-also --no-drive 2>$O/b.err | python -c "import json,sys; d=json.load(sys.stdin); print('FAKE16=$v', d['ms_per_step'])" | tee -a $O/summary.txt; done
for v in 0 1; do LP_VGG_FAKE16=$v timeout 300 python bench.py --workload finetune_step --steps 40 --warmup 10 --no-cpu-baseline --no-also --no-drive 2>$O/b.err | python -c "import json,sys; d=json.load(sys.stdin); print('finetune FAKE16=$v', d['ms_per_step'])" | tee -a $O/summary.txt; done
