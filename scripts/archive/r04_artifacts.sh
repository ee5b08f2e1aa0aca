#!/bin/bash
# Regenerates the judged artifacts of round 4 on the GPU box (outputs under gpurun_out/r04a/, copied to profiles/ afterwards):
#   * the default bench line (meta-training step = the 1/2/4/8-GPU workload; identity encoder bf16x3 head + fp16 tail; strict-mode and
#     fine-tuning side lines; cpu_baseline with per-NUMA-node pinned workers at bs 8) + the per-shape conv table,
#   * rocprofv3 --kernel-trace --stats of the same command + one-step kernel breakdowns (meta-training and fine-tuning),
#   * PMC passes (separate --pmc runs, --kernel-trace only, counters restricted by --kernel-include-regex) over the launch population of the
#     META-TRAINING step: HBM traffic (FETCH_SIZE, WRITE_SIZE) and MFMA utilisation of the 3x3 conv kernels, the 1x1 convs, the weight-
#     gradient kernels and the BatchNorm-backward kernels,
#   * the in-graph roofline figure of the 3x3 family, conv micro-benchmarks, FSTH_plus 512 / generator lines, the 2-rank functional run
#     started by `python bench.py --gpus 2` itself, the parity JSONs of tests/test_metatrain_full_gpu.py.
O=$GRAFT_REPO_ROOT/gpurun_out/r04a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=r04
LP_PARITY_OUT=$O timeout 600 python -m pytest tests/test_metatrain_full_gpu.py -m gpu -q -s > $O/${R}_parity_configs2.log 2>&1; echo "parity configs2 rc=$?" | tee $O/summary.txt
cp $O/${R}_parity_configs2_*.json profiles/ 2>/dev/null      # bench.py reads them (this run's copy; the files are committed afterwards)
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
FAM="conv_pipe_kernel|conv_dma_kernel|conv_wgrad_kernel|wgrad1x1_kernel|bn_bwd16"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES"; do
  tag=$(echo $c | tr ' ' '+')
  timeout 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "$FAM" --output-format csv -d $O/${R}_pmc_$tag -o ${R} -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also --no-drive > $O/${R}_pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?" >> $O/summary.txt
  rm -f $O/${R}_pmc_$tag/${R}_kernel_trace.csv
done
python scripts/pmc_summary.py --json "conv_pipe_kernel|conv_dma_kernel<3" $O/${R}_pmc_*/*counter_collection.csv > $O/${R}_pmc_conv3x3_metatrain.json 2> $O/${R}_pmc_summary.err
python scripts/pmc_summary.py --json "conv_dma_kernel<1" $O/${R}_pmc_*/*counter_collection.csv > $O/${R}_pmc_conv1x1_metatrain.json 2>> $O/${R}_pmc_summary.err
python scripts/pmc_summary.py --json "conv_wgrad_kernel" $O/${R}_pmc_*/*counter_collection.csv > $O/${R}_pmc_conv_wgrad_metatrain.json 2>> $O/${R}_pmc_summary.err
python scripts/pmc_summary.py --json "wgrad1x1_kernel" $O/${R}_pmc_*/*counter_collection.csv > $O/${R}_pmc_wgrad1x1_metatrain.json 2>> $O/${R}_pmc_summary.err
python scripts/pmc_summary.py --json "bn_bwd16" $O/${R}_pmc_*/*counter_collection.csv > $O/${R}_pmc_bn_bwd16_metatrain.json 2>> $O/${R}_pmc_summary.err
python scripts/pmc_summary.py $O/${R}_pmc_*/*counter_collection.csv > $O/${R}_pmc_metatrain_step.csv 2>> $O/${R}_pmc_summary.err
rm -rf $O/${R}_pmc_*/
timeout 300 python bench.py --workload generator --generator FSTH_plus --image_size 512 --batch 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/${R}_bench_fsthplus512.json 2> $O/${R}_bench_fsthplus512.err
timeout 300 python bench.py --workload generator --steps 20 --warmup 5 --no-cpu-baseline > $O/${R}_bench_generator.json 2> $O/${R}_bench_generator.err
env -u WORLD_SIZE -u RANK -u LOCAL_RANK timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --backend gloo --no-cpu-baseline --no-also --no-drive > $O/${R}_bench_dp2_gloo_one_gpu_functional.json 2> $O/${R}_bench_dp2.err
PREC=2 WHAT=conv,wgrad timeout 120 python scripts/conv_micro.py > $O/${R}_conv_micro_f16.txt 2>&1
LP_OVERLAP=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/${R}_bench_one_stream.json 2> $O/${R}_bench_one_stream.err
cut -c1-3000 $O/${R}_bench.json
tail -3 $O/${R}_bench.err
cat $O/${R}_pmc_conv3x3_metatrain.json; cat $O/${R}_conv3x3_in_graph.json
cat $O/summary.txt
head -30 $O/${R}_step_breakdown_metatrain.csv
cut -c1-400 $O/${R}_bench_dp2_gloo_one_gpu_functional.json; tail -2 $O/${R}_bench_dp2.err
