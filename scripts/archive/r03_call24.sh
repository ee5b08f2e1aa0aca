#!/bin/bash
# round 3, GPU call 24: convergence comparison f16 vs bf16x3 (fine-tuning 300 steps, meta-training 200 steps); default meta-training step re-measured
O=$GRAFT_REPO_ROOT/gpurun_out/r03c24
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('metatrain default', j['value'], j['ms_per_step'])"
timeout 600 python scripts/convergence_compare.py finetune 300 4 > $O/r03_convergence_finetune.txt 2>&1
timeout 900 python scripts/convergence_compare.py metatrain 200 4 > $O/r03_convergence_metatrain.txt 2>&1
cut -c1-400 $O/r03_convergence_finetune.txt | tail -12
cut -c1-400 $O/r03_convergence_metatrain.txt | tail -12
