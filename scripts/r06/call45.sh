#!/bin/bash
# round 6, call 45: the identity encoder's weight packs on a side stream beside the stem's im2col pass (LP_E_PACKS_SIDE): tests + A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06y; mkdir -p $O
timeout 1200 python -m pytest tests/test_resnext_hip.py tests/test_e1_full_gpu.py tests/test_metatrain_step.py tests/test_streams_gpu.py tests/test_train_entry_gpu.py tests/test_data_parallel_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
for i in 1 2 3 4; do for k in 1 0; do
  LP_E_PACKS_SIDE=$k python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b$k.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('metatrain packs_side=$k', d['ms_per_step'])" | tee -a $O/ab.txt
done; done
tail -3 $O/b1.err
