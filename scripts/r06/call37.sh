#!/bin/bash
# round 6, call 37 (second run: + planes-only downsample form): identity shortcuts of the bf16x3 encoder blocks from the operand planes (lp_bn_add_act_planes, LP_E_RES16): tests + A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06q; mkdir -p $O
timeout 900 python -m pytest tests/test_resnext_hip.py tests/test_abi.py tests/test_e1_full_gpu.py tests/test_metatrain_step.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.txt
timeout 900 python -m pytest tests/test_metatrain_full_gpu.py -x -q -m gpu -s 2>&1 | grep -E "parity-configs2|passed|failed" | cut -c1-1500 | tee -a $O/tests.txt
for i in 1 2 3; do for k in 1 0; do
  LP_E_RES16=$k python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b$k.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('metatrain res16=$k', d['ms_per_step'])" | tee -a $O/ab.txt
done; done
