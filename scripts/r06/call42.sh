#!/bin/bash
# round 6, call 42: depthwise 3x3 weight gradient with row-segment items + sliding window, parallel fixed-order reduce: tests, kernel stats, A/B against the previous library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06v; mkdir -p $O
timeout 900 python -m pytest tests/test_mobilenet_train_hip.py tests/test_e2_full_gpu.py tests/test_metatrain_step.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
for i in 1 2 3; do for k in new prev; do
  if [ $k = prev ]; then export LP_LIB_OVERRIDE=$GRAFT_REPO_ROOT/probes/liblp_hip_prev.so; else unset LP_LIB_OVERRIDE; fi
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b$k.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('metatrain lib=$k', d['ms_per_step'])" | tee -a $O/ab.txt
done; done
unset LP_LIB_OVERRIDE
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r06 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-drive > $O/prof.log 2>&1
grep -E "dwconv3x3" $O/prof/r06_kernel_stats.csv | cut -c1-60,200-330 | tee $O/dw_stats.txt
grep -E "dwconv3x3" $O/prof/r06_kernel_stats.csv | awk -F'","' '{print $1, $2, $3, $4}' | cut -c1-200 | tee -a $O/dw_stats.txt
rm -rf $O/prof
