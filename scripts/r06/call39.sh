#!/bin/bash
# round 6, call 39: kernel trace of the meta-training step -> one-step breakdown (marker period read off the trace) + in-graph 3x3 figure
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06s; mkdir -p $O; R=r06
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive --shapes $O/${R}_conv_shapes_metatrain.csv > $O/${R}_bench_quick.json 2> $O/${R}_bench_quick.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_meta -o ${R} -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-drive > $O/${R}_prof_meta.log 2>&1
python scripts/step_breakdown.py $O/${R}_prof_meta/${R}_kernel_trace.csv > $O/${R}_step_breakdown_metatrain.csv 2> $O/breakdown.err
python scripts/trace_concurrency.py $O/${R}_prof_meta/${R}_kernel_trace.csv > $O/concurrency.txt 2>&1
python scripts/in_graph_conv.py $O/${R}_step_breakdown_metatrain.csv $O/${R}_conv_shapes_metatrain.csv > $O/${R}_conv3x3_in_graph.json 2>> $O/breakdown.err
rm -rf $O/${R}_prof_meta
cat $O/breakdown.err; head -12 $O/${R}_step_breakdown_metatrain.csv; cat $O/${R}_conv3x3_in_graph.json | cut -c1-600; tail -5 $O/concurrency.txt
