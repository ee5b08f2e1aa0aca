#!/bin/bash
# round 5, call 40: BASELINE.md section-3 CPU protocol (3 warm-up + 10 timed) on the box's host cores: meta-training step, generator-only, drive frame
O=$GRAFT_REPO_ROOT/gpurun_out/r05af
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python bench.py --cpu-baseline-only --cpu-baseline-full > $O/r05_cpu_baseline_full.json 2> $O/r05_cpu_baseline_full.err; echo "cpu full rc=$?"
cut -c1-1500 $O/r05_cpu_baseline_full.json
