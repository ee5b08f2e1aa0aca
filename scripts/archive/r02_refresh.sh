#!/bin/bash
# Re-measures the bench lines and one-step breakdowns of profiles/r02_* after a host-side or elementwise-kernel change (the per-layer
# micro-benchmarks and PMC passes of scripts/r02_artifacts.sh are not repeated).  Outputs under gpurun_out/r02b/.
O=$GRAFT_REPO_ROOT/gpurun_out/r02b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=r02
timeout 600 python bench.py --shapes $O/${R}_conv_shapes_f16.csv > $O/${R}_bench_f16.json 2> $O/${R}_bench_f16.err
timeout 300 python bench.py --prec bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --shapes $O/${R}_conv_shapes_bf16x3.csv > $O/${R}_bench_bf16x3.json 2> $O/${R}_bench_bf16x3.err
timeout 400 python bench.py --workload metatrain_step --steps 20 --warmup 5 --no-cpu-baseline > $O/${R}_bench_metatrain_f16.json 2> $O/${R}_bench_metatrain_f16.err
for P in f16 bf16x3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_$P -o ${R} -- python bench.py --prec $P --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/${R}_prof_$P.log 2>&1
  python scripts/step_breakdown.py $O/${R}_prof_$P/${R}_kernel_trace.csv > $O/${R}_step_breakdown_$P.csv 2>> $O/${R}_prof_$P.log
  rm -f $O/${R}_prof_$P/${R}_kernel_trace.csv
done
head -4 $O/${R}_step_breakdown_f16.csv
