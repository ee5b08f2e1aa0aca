#!/bin/bash
# Regenerates the judged artifacts of a round on the GPU box: bench lines, rocprofv3 kernel stats of the same command, PMC passes.
R=${1:-r01}
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py > $O/${R}_bench_bf16x3.json 2> $O/${R}_bench_bf16x3.err
python bench.py --prec bf16 --no-cpu-baseline > $O/${R}_bench_bf16.json 2> $O/${R}_bench_bf16.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_x3 -o ${R} -- python bench.py --no-cpu-baseline > $O/${R}_prof_x3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_bf16 -o ${R} -- python bench.py --prec bf16 --no-cpu-baseline > $O/${R}_prof_bf16.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${R}_pmc_fetch -o ${R} -- python bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline > $O/${R}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${R}_pmc_write -o ${R} -- python bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline > $O/${R}_pmc_write.log 2>&1
rm -f $O/${R}_prof_*/*kernel_trace.csv $O/${R}_pmc_*/*kernel_trace.csv   # large; the stats/counter tables are what is kept
python scripts/pmc_summary.py $O/${R}_pmc_fetch/*counter_collection.csv $O/${R}_pmc_write/*counter_collection.csv > $O/${R}_pmc_summary.csv 2>> $O/${R}_pmc_fetch.log
ls -la $O/${R}_p* | head -40
cat $O/${R}_bench_bf16x3.json $O/${R}_bench_bf16.json
