#!/bin/bash
# round 6, call 24: concurrency picture of the captured meta-training step (which kernels run alone = the critical path), final artifact tree
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o r06 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-drive > $O/prof.log 2>&1
T=$(ls $O/prof/*/r06_kernel_trace.csv $O/prof/r06_kernel_trace.csv 2>/dev/null | head -1)
python scripts/trace_concurrency.py $T 60 > $O/concurrency.csv 2> $O/conc.err
head -70 $O/concurrency.csv | cut -c1-200
rm -rf $O/prof
