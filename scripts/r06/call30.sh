#!/bin/bash
# round 6, call 30: remaining optional branches on the fine-tuning step (default branches on)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06i; mkdir -p $O
run() { env "$@" python bench.py --workload finetune_step --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'])" | tee -a $O/ft.txt; }
run A=0
run LP_OVERLAP_TARGETS=1
run LP_OVERLAP_TARGETS=2
run LP_OVERLAP_OPTIMIZER=1
run A=1
