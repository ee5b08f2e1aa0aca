#!/bin/bash
# round 4, GPU call 17: what is left on conv_dma_kernel<3> (grouped convs of the identity encoder, small maps): ping-pong vs single-group schedule on the step
O=$GRAFT_REPO_ROOT/gpurun_out/r04c17
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "A=1" "LP_CONV_PP=0" "A=1" "LP_CONV_PP=0"; do
  env $v timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/b.json 2> $O/b.err
  echo "$v"; python -c "
import json; j=json.load(open('$O/b.json')); print('   ', j['ms_per_step'], 'ms', j['value'], 'img/s', 'gconv', j['roofline_gconv']['achieved'], 'GB/s avg', j['roofline_gconv']['avg_launch_us'])"
done
