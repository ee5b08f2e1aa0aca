#!/bin/bash
# round 6, call 36: fp16 operands in the LAST k blocks of the bf16x3 generator (LP_G_F16_TAIL=k): tie-masked gradient figures at 256 x 256 + step time
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06o; mkdir -p $O
timeout 600 python -m pytest tests/test_generator_module.py tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
for k in 0 1 2 3; do
  LP_G_F16_TAIL=$k timeout 600 python -m pytest "tests/test_full_size_parity.py::test_generator_256_vs_oracle[default-0.001-0.001]" -x -q -m gpu -s 2>&1 | grep -E "parity-256|passed|failed|Error" | sed "s/^/tail=$k /" | tee -a $O/parity.txt
done
for i in 1 2; do for k in 0 2 3 1; do
  LP_G_F16_TAIL=$k python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b$k.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('metatrain tail=$k', d['ms_per_step'])" | tee -a $O/ab.txt
done; done
