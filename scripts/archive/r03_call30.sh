#!/bin/bash
# round 3, GPU call 30: spectral-norm power iterations + weight packs of G and D on a side stream beside the encoders: parity + A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r03c30
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_streams_gpu.py tests/test_metatrain_step.py tests/test_train_step.py tests/test_generator_module.py tests/test_discriminator_criterions.py tests/test_data_parallel_gpu.py tests/test_train_entry_gpu.py tests/test_checkpoint_fixture.py -m gpu -q -s > $O/tests.log 2>&1
echo "tests rc=$?" | tee $O/summary.txt
grep -E "\[streams\]|passed|failed" $O/tests.log | cut -c1-300 | tail -6
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-300
{
for v in 1 0 1 0; do
LP_OVERLAP_PREPARE=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('metatrain, SN power iterations + weight packs of G and D beside the encoders=$v', j['value'], j['ms_per_step'])"
done
} 2>&1 | tee $O/r03_prepare_ahead.txt
