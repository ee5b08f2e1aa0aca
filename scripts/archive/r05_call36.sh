#!/bin/bash
# round 5, call 36: kernel statistics of the reflection-padding step
O=$GRAFT_REPO_ROOT/gpurun_out/r05ad
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o refl -- python $GRAFT_REPO_ROOT/bench.py --padding reflection --steps 3 --warmup 2 --eager --no-cpu-baseline --no-also --no-drive > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -40 "$f" | cut -c1-220 | tee $O/kernel_stats_top.txt
rm -rf $O/prof
