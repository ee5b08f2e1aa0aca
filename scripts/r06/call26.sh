#!/bin/bash
# round 6, call 26: lp_pool_grad_pack (the fused backward of nn.ConvPoolFn): kernel + module parity, then the step with / without it
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06e; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_conv_pool.py tests/test_discriminator_criterions.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.txt
timeout 900 python -m pytest tests/test_full_size_parity.py tests/test_streams_gpu.py tests/test_train_entry_gpu.py -x -q -m gpu 2>&1 | tail -5 >> $O/tests.txt
cat $O/tests.txt
for i in 1 2; do
for f in 1 0; do
  LP_POOL_GRAD_FUSED=$f python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b$f.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused=$f', d['ms_per_step'])" | tee -a $O/ab.txt
done; done
