#!/bin/bash
# round 6, call 28: stream branches on the FINE-TUNING step, re-measured on the round-6 precision assignment (bf16x3 generator / critic tail)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06g; mkdir -p $O
run() { env "$@" python bench.py --workload finetune_step --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d['config']['streams']['concurrent_branches'])" | tee -a $O/ft.txt; }
run A=0
run LP_OVERLAP_DPASSES=1
run LP_OVERLAP_DPASSES=1 LP_OVERLAP_PREPARE=1
run LP_OVERLAP_DPASSES=1 LP_OVERLAP_PREPARE=1 LP_OVERLAP_REAL=1
run LP_OVERLAP_DPASSES=1 LP_OVERLAP_PREPARE=1 LP_OVERLAP_REAL=1 LP_OVERLAP_CRITERIONS=1
run LP_OVERLAP_CRITERIONS=1
run A=1
