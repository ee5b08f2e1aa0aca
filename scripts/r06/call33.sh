#!/bin/bash
# round 6, call 33: BatchNorm-backward pass 1 from the data-gradient epilogue (lp_conv16_dgrad_bnb, ABI 12): parity, then the step with / without it
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06l; mkdir -p $O
timeout 1200 python -m pytest tests/test_resnext_hip.py tests/test_hip_ops.py tests/test_mobilenet_train_hip.py -x -q -m gpu -s 2>&1 | grep -E "^\[bnb\]|passed|failed|Error|error" | tail -12 > $O/tests.txt
cat $O/tests.txt
timeout 1200 python -m pytest tests/test_e1_full_gpu.py tests/test_metatrain_step.py tests/test_streams_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O/tests.txt
for i in 1 2 3; do for f in 1 0; do
  LP_E_BNB_FUSE=$f python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b$f.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse=$f', d['ms_per_step'])" | tee -a $O/ab.txt
done; done
