#!/bin/bash
# round 4, GPU call 27: fresh one-step breakdown + conv shape table of the current tree
O=$GRAFT_REPO_ROOT/gpurun_out/r04c27
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=r04
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive --shapes $O/${R}_conv_shapes_metatrain.csv > $O/${R}_bench_quick.json 2> $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_meta -o ${R} -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-drive > $O/${R}_prof_meta.log 2>&1
python scripts/step_breakdown.py $O/${R}_prof_meta/${R}_kernel_trace.csv > $O/${R}_step_breakdown_metatrain.csv 2>> $O/${R}_prof_meta.log
cp $O/${R}_prof_meta/${R}_kernel_stats.csv $O/${R}_metatrain_step_kernel_stats.csv 2>/dev/null
rm -f $O/${R}_prof_meta/${R}_kernel_trace.csv
python scripts/in_graph_conv.py $O/${R}_step_breakdown_metatrain.csv $O/${R}_conv_shapes_metatrain.csv > $O/${R}_conv3x3_in_graph.json 2>> $O/${R}_prof_meta.log
head -45 $O/${R}_step_breakdown_metatrain.csv | cut -c1-150; cat $O/${R}_conv3x3_in_graph.json
