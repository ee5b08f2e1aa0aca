#!/bin/bash
# round 6, call 35: fine-tuning step with optimizer_G.step + EMA beside loss_D.backward by default: parity (graph == eager, goldens) + bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06n; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_entry_gpu.py tests/test_train_step.py tests/test_streams_gpu.py tests/test_prefetch.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
for i in 1 2; do for f in 1 0; do
  LP_OVERLAP_OPTIMIZER=$f python bench.py --workload finetune_step --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b$f.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('finetune optimizer=$f', d['ms_per_step'], d['config']['streams']['concurrent_branches'])" | tee -a $O/ab.txt
done; done
