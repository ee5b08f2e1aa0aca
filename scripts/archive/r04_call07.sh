#!/bin/bash
# round 4, GPU call 7: where does the captured step with the encoders' backward beside loss_D.backward crash?
O=$GRAFT_REPO_ROOT/gpurun_out/r04c07
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LP_OVERLAP_EBWD=1 timeout 200 python -X faulthandler bench.py --image_size 128 --steps 5 --warmup 2 --no-cpu-baseline --no-also --no-drive > $O/bench128.json 2> $O/bench128.err
echo "rc=$?"; tail -40 $O/bench128.err | cut -c1-300; cut -c1-300 $O/bench128.json
