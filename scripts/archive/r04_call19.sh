#!/bin/bash
# round 4, GPU call 19: DMA-staged 3x3 weight-gradient kernel (wgrad3_pipe_kernel): parity cases (forced / default / off) + micro A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r04c19
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_wgrad3_pipe.py -m gpu -q -x -s > $O/tests.log 2>&1; echo "wgrad3 tests rc=$?" | tee $O/summary.txt
grep -E "passed|failed|FAILED|wgrad3_pipe" $O/tests.log | cut -c1-220 | tail -40
for v in "LP_WGRAD3_PIPE=0" "LP_WGRAD3_PIPE=1" "LP_WGRAD3_PIPE=1 LP_WGRAD3_WGS=512" "LP_WGRAD3_PIPE=1 LP_WGRAD3_WGS=2048"; do
  echo "== $v" >> $O/micro.log
  env $v SHAPES=wgrad BIAS=1 PREC=2 WHAT=wgrad REPS=30 timeout 120 python scripts/conv_micro.py 2>&1 | grep -v amdgpu.ids >> $O/micro.log
done
cat $O/micro.log
