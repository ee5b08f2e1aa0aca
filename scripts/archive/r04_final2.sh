#!/bin/bash
# round 4, last artifact call on the final tree: parity JSONs of the full configs[2] forward, the default bench line (+ shapes), one-step breakdown, in-graph figure
O=$GRAFT_REPO_ROOT/gpurun_out/r04g
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=r04
LP_PARITY_OUT=$O timeout 600 python -m pytest tests/test_metatrain_full_gpu.py tests/test_train_entry_gpu.py -m gpu -q -s > $O/${R}_parity_configs2.log 2>&1; echo "parity + entry tests rc=$?" | tee $O/summary.txt
grep -E "parity-configs2|passed|failed" $O/${R}_parity_configs2.log | cut -c1-600 | tail -4
cp $O/${R}_parity_configs2_*.json profiles/ 2>/dev/null
timeout 1500 python bench.py --shapes $O/${R}_conv_shapes_metatrain.csv > $O/${R}_bench.json 2> $O/${R}_bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_meta -o ${R} -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-drive > $O/${R}_prof_meta.log 2>&1
python scripts/step_breakdown.py $O/${R}_prof_meta/${R}_kernel_trace.csv > $O/${R}_step_breakdown_metatrain.csv 2>> $O/${R}_prof_meta.log
cp $O/${R}_prof_meta/${R}_kernel_stats.csv $O/${R}_metatrain_step_kernel_stats.csv 2>/dev/null
rm -f $O/${R}_prof_meta/${R}_kernel_trace.csv
python scripts/in_graph_conv.py $O/${R}_step_breakdown_metatrain.csv $O/${R}_conv_shapes_metatrain.csv > $O/${R}_conv3x3_in_graph.json 2>> $O/${R}_prof_meta.log
timeout 300 python bench.py --prec bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/${R}_bench_bf16x3.json 2> $O/${R}_bench_bf16x3.err
cut -c1-300 $O/${R}_bench.json; echo; cat $O/${R}_conv3x3_in_graph.json; cat $O/summary.txt
