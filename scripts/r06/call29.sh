#!/bin/bash
# round 6, call 29: fine-tuning step with its stream branches ON by default: parity tests + bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06h; mkdir -p $O
timeout 1200 python -m pytest tests/test_streams_gpu.py tests/test_train_step.py tests/test_train_entry_gpu.py tests/test_optim.py tests/test_prefetch.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -12 > $O/tests.txt
cat $O/tests.txt
for i in 1 2; do python bench.py --workload finetune_step --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('finetune default', d['ms_per_step'], d['config']['streams']['concurrent_branches'])" | tee -a $O/ft.txt; done
LP_OVERLAP=0 python bench.py --workload finetune_step --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('finetune one stream', d['ms_per_step'])" | tee -a $O/ft.txt
