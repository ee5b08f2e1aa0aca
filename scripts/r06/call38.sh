#!/bin/bash
# round 6, call 38: the generator's weight gradients deferred to the critic-backward stream (LP_OVERLAP_GWGRAD): parity test + A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06r; mkdir -p $O
timeout 900 python -m pytest tests/test_streams_gpu.py -x -q -m gpu -s 2>&1 | grep -E "^\[streams\]|passed|failed|Error" | cut -c1-600 | tee $O/tests.txt
LP_OVERLAP_GWGRAD=1 timeout 900 python -m pytest tests/test_train_entry_gpu.py tests/test_metatrain_step.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O/tests.txt
for i in 1 2 3; do for k in 1 0; do
  LP_OVERLAP_GWGRAD=$k python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b$k.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('metatrain gwgrad=$k', d['ms_per_step'], d['config']['streams']['concurrent_branches'])" | tee -a $O/ab.txt
done; done
tail -5 $O/b1.err
