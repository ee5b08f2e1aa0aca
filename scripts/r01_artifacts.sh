#!/bin/bash
# Regenerates the judged artifacts of a round on the GPU box: bench lines, rocprofv3 kernel stats of the same command,
# a per-step kernel breakdown, and the PMC (HBM traffic) passes of the dominant kernel on the conv micro-benchmark.
# (PMC passes over the whole training step were tried and hang rocprofv3 on this pool -- they are not run.)
R=${1:-r01}
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --shapes $O/${R}_conv_shapes_bf16x3.csv > $O/${R}_bench_bf16x3.json 2> $O/${R}_bench_bf16x3.err
timeout 300 python bench.py --prec bf16 --no-cpu-baseline --shapes $O/${R}_conv_shapes_bf16.csv > $O/${R}_bench_bf16.json 2> $O/${R}_bench_bf16.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for P in bf16x3 bf16; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_$P -o ${R} -- python bench.py --prec $P --no-cpu-baseline > $O/${R}_prof_$P.log 2>&1
  python scripts/step_breakdown.py $O/${R}_prof_$P/${R}_kernel_trace.csv > $O/${R}_step_breakdown_$P.csv 2>> $O/${R}_prof_$P.log
  rm -f $O/${R}_prof_$P/${R}_kernel_trace.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  PREC=1 REPS=2 timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${R}_pmc_conv_$c -o ${R} -- python scripts/conv_micro.py > $O/${R}_pmc_conv_$c.log 2>&1
  rm -f $O/${R}_pmc_conv_$c/${R}_kernel_trace.csv
done
python scripts/pmc_summary.py $O/${R}_pmc_conv_*/*counter_collection.csv > $O/${R}_pmc_conv_summary.csv
cat $O/${R}_bench_bf16x3.json $O/${R}_bench_bf16.json
head -30 $O/${R}_step_breakdown_bf16x3.csv
