#!/bin/bash
# round 4, GPU call 5: does a strict first (image) layer of the critic fix the fp16 score error at the full geometry?
O=$GRAFT_REPO_ROOT/gpurun_out/r04c05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
  LP_D_RGB_STRICT=$v timeout 300 python tests/test_metatrain_full_gpu.py $O/full_rgbstrict$v.json > $O/full_rgbstrict$v.log 2>&1
  python - <<PY
import json
try:
    d=json.load(open('$O/full_rgbstrict$v.json')); print('LP_D_RGB_STRICT=$v', {k: f'{x:.2e}' for k,x in sorted(d['errors'].items(), key=lambda kv:-kv[1])})
except Exception as e: print('failed $v', e)
PY
  tail -3 $O/full_rgbstrict$v.log | cut -c1-300
done
LP_D_RGB_STRICT=1 timeout 300 python -m pytest tests/test_discriminator_criterions.py tests/test_full_size_parity.py -m gpu -q -x -k "discriminator or critic" > $O/dtests.log 2>&1; echo "D tests rc=$?"
tail -3 $O/dtests.log
