#!/bin/bash
# round 6, call 46: the tree at the start of this session (commit 567049d, materialised under _r06_start/ for this call only) against the final tree, alternating on one box
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06z; mkdir -p $O
for i in 1 2 3 4; do for k in final start; do
  if [ $k = start ]; then D=$GRAFT_REPO_ROOT/_r06_start; else D=$GRAFT_REPO_ROOT; fi
  (cd $D && python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b$k.err) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('metatrain tree=$k', d['ms_per_step'])" | tee -a $O/ab.txt
done; done
for i in 1 2; do for k in final start; do
  if [ $k = start ]; then D=$GRAFT_REPO_ROOT/_r06_start; else D=$GRAFT_REPO_ROOT; fi
  (cd $D && python bench.py --workload finetune_step --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2>> $O/b$k.err) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('finetune tree=$k', d['ms_per_step'])" | tee -a $O/ab.txt
done; done
tail -3 $O/bstart.err
