#!/bin/bash
# round 4, final artifact call (tree with the DMA-staged weight gradients, grouped diagonal forms, target-image overlap, 16-bit target taps):
# parity JSONs, bench lines, one-step breakdowns, in-graph 3x3 figure, PMC passes per kernel family
O=$GRAFT_REPO_ROOT/gpurun_out/r04h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=r04
LP_PARITY_OUT=$O timeout 600 python -m pytest tests/test_metatrain_full_gpu.py tests/test_train_entry_gpu.py -m gpu -q -s > $O/${R}_parity_configs2.log 2>&1; echo "parity + entry tests rc=$?" | tee $O/summary.txt
grep -E "parity-configs2|passed|failed" $O/${R}_parity_configs2.log | cut -c1-400 | tail -4
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
FAM="conv_pipe_kernel|conv_dma_kernel|conv_wgrad_kernel|wgrad3_pipe_kernel|wgrad1x1_kernel|bn_bwd16"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES"; do
  tag=$(echo $c | tr ' ' '+')
  timeout 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "$FAM" --output-format csv -d $O/${R}_pmc_$tag -o ${R} -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also --no-drive > $O/${R}_pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?" >> $O/summary.txt
  rm -f $O/${R}_pmc_$tag/${R}_kernel_trace.csv
done
python scripts/pmc_summary.py --json "conv_pipe_kernel|conv_dma_kernel<3" $O/${R}_pmc_*/*counter_collection.csv > $O/${R}_pmc_conv3x3_metatrain.json 2> $O/${R}_pmc_summary.err
python scripts/pmc_summary.py --json "conv_dma_kernel<1" $O/${R}_pmc_*/*counter_collection.csv > $O/${R}_pmc_conv1x1_metatrain.json 2>> $O/${R}_pmc_summary.err
python scripts/pmc_summary.py --json "wgrad3_pipe_kernel|conv_wgrad_kernel" $O/${R}_pmc_*/*counter_collection.csv > $O/${R}_pmc_conv_wgrad_metatrain.json 2>> $O/${R}_pmc_summary.err
python scripts/pmc_summary.py --json "wgrad1x1_kernel" $O/${R}_pmc_*/*counter_collection.csv > $O/${R}_pmc_wgrad1x1_metatrain.json 2>> $O/${R}_pmc_summary.err
python scripts/pmc_summary.py --json "bn_bwd16" $O/${R}_pmc_*/*counter_collection.csv > $O/${R}_pmc_bn_bwd16_metatrain.json 2>> $O/${R}_pmc_summary.err
python scripts/pmc_summary.py $O/${R}_pmc_*/*counter_collection.csv > $O/${R}_pmc_metatrain_step.csv 2>> $O/${R}_pmc_summary.err
rm -rf $O/${R}_pmc_*/
timeout 300 python bench.py --prec bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/${R}_bench_bf16x3.json 2> $O/${R}_bench_bf16x3.err
LP_OVERLAP=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/${R}_bench_one_stream.json 2> $O/${R}_bench_one_stream.err
SHAPES=wgrad BIAS=1 PREC=2 WHAT=wgrad REPS=30 timeout 120 python scripts/conv_micro.py 2>&1 | grep -v amdgpu.ids > $O/${R}_wgrad3_micro_new.txt
LP_WGRAD3_PIPE=0 SHAPES=wgrad BIAS=1 PREC=2 WHAT=wgrad REPS=30 timeout 120 python scripts/conv_micro.py 2>&1 | grep -v amdgpu.ids > $O/${R}_wgrad3_micro_old.txt
cut -c1-300 $O/${R}_bench.json; echo; cat $O/${R}_conv3x3_in_graph.json; cat $O/${R}_pmc_conv_wgrad_metatrain.json | cut -c1-600; cat $O/summary.txt
