#!/bin/bash
# round 6, call 34: optimizer_D.step, the generator's slice of optimizer_G and the generator's EMA beside the encoders' backward (LP_OVERLAP_EARLY): parity + A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06m; mkdir -p $O
timeout 1500 python -m pytest tests/test_optim.py tests/test_streams_gpu.py tests/test_train_entry_gpu.py tests/test_metatrain_step.py tests/test_train_step.py tests/test_data_parallel_gpu.py tests/test_checkpoint_fixture.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.txt
for i in 1 2 3; do for f in 1 0; do
  LP_OVERLAP_EARLY=$f python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b$f.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('early=$f', d['ms_per_step'])" | tee -a $O/ab.txt
done; done
