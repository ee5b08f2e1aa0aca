#!/bin/bash
# round 4, final GPU call: the whole GPU suite + smoke on the final tree, the default bench line (+ per-shape table), the one-step kernel
# breakdowns and the in-graph 3x3 figure, and the BASELINE.md section-3 CPU protocol (>= 3 warm-up + >= 10 timed steps) of the meta-training step
O=$GRAFT_REPO_ROOT/gpurun_out/r04f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=r04
timeout 1500 python -m pytest tests -m gpu -q > $O/${R}_pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee $O/summary.txt
tail -4 $O/${R}_pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${R}_smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 1500 python bench.py --shapes $O/${R}_conv_shapes_metatrain.csv > $O/${R}_bench.json 2> $O/${R}_bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_meta -o ${R} -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-drive > $O/${R}_prof_meta.log 2>&1
python scripts/step_breakdown.py $O/${R}_prof_meta/${R}_kernel_trace.csv > $O/${R}_step_breakdown_metatrain.csv 2>> $O/${R}_prof_meta.log
cp $O/${R}_prof_meta/${R}_kernel_stats.csv $O/${R}_metatrain_step_kernel_stats.csv 2>/dev/null
rm -f $O/${R}_prof_meta/${R}_kernel_trace.csv
python scripts/in_graph_conv.py $O/${R}_step_breakdown_metatrain.csv $O/${R}_conv_shapes_metatrain.csv > $O/${R}_conv3x3_in_graph.json 2>> $O/${R}_prof_meta.log
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_ft -o ${R} -- python bench.py --workload finetune_step --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-drive > $O/${R}_prof_ft.log 2>&1
python scripts/step_breakdown.py $O/${R}_prof_ft/${R}_kernel_trace.csv > $O/${R}_step_breakdown_finetune.csv 2>> $O/${R}_prof_ft.log
cp $O/${R}_prof_ft/${R}_kernel_stats.csv $O/${R}_finetune_step_kernel_stats.csv 2>/dev/null
rm -f $O/${R}_prof_ft/${R}_kernel_trace.csv
timeout 1200 python bench.py --cpu-baseline-only --cpu-baseline-full > $O/${R}_cpu_baseline_full.json 2> $O/${R}_cpu_baseline_full.err; echo "cpu full rc=$?" | tee -a $O/summary.txt
cut -c1-1200 $O/${R}_bench.json; echo
cat $O/${R}_conv3x3_in_graph.json
cut -c1-700 $O/${R}_cpu_baseline_full.json; echo
cat $O/summary.txt
head -14 $O/${R}_step_breakdown_metatrain.csv
