#!/bin/bash
# round 3, GPU call 25: convergence comparison with a chaos-floor control run; default meta-training step re-measured twice
O=$GRAFT_REPO_ROOT/gpurun_out/r03c25
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('metatrain default', j['value'], j['ms_per_step'])"; done
LP_OVERLAP=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('metatrain one stream', j['value'], j['ms_per_step'])"
timeout 900 python scripts/convergence_compare.py metatrain 200 4 > $O/r03_convergence_metatrain.txt 2>&1
timeout 600 python scripts/convergence_compare.py finetune 300 4 > $O/r03_convergence_finetune.txt 2>&1
cut -c1-250 $O/r03_convergence_metatrain.txt | tail -70
grep "^# " $O/r03_convergence_finetune.txt | cut -c1-250
