#!/bin/bash
# round 6, call 44: workgroup-per-channel statistics finalize for >= 256 partials: tests, A/B against the previous library, kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06x; mkdir -p $O
timeout 900 python -m pytest tests/test_resnext_hip.py tests/test_conv_stats.py tests/test_hip_ops.py tests/test_e1_full_gpu.py tests/test_metatrain_step.py tests/test_mobilenet_train_hip.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
for i in 1 2 3 4; do for k in new prev; do
  if [ $k = prev ]; then export LP_LIB_OVERRIDE=$GRAFT_REPO_ROOT/probes/liblp_hip_prev.so; else unset LP_LIB_OVERRIDE; fi
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b$k.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('metatrain lib=$k', d['ms_per_step'])" | tee -a $O/ab.txt
done; done
unset LP_LIB_OVERRIDE
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r06 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-drive > $O/prof.log 2>&1
grep -E "instnorm_finalize" $O/prof/r06_kernel_stats.csv | awk -F'","' '{print $1, $2, $3, $4, $6, $7}' | cut -c1-260 | tee $O/stats.txt
rm -rf $O/prof
