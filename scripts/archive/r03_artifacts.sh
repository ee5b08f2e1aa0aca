#!/bin/bash
# Regenerates the judged artifacts of round 3 on the GPU box (outputs under gpurun_out/r03a/, copied to profiles/ afterwards):
#   * the default bench line (meta-training step = the 1/2/4/8-GPU workload, with the fine-tuning and strict-mode side lines and cpu_baseline),
#   * rocprofv3 --kernel-trace --stats of the same command + one-step kernel breakdowns (meta-training and fine-tuning),
#   * PMC passes (separate --pmc runs, --kernel-trace only, counters restricted to the conv kernels by --kernel-include-regex) over the
#     launch population of one generator fwd+bwd step: HBM traffic (FETCH_SIZE, WRITE_SIZE) and MFMA utilisation of conv_dma_kernel,
#   * per-layer conv micro-benchmarks, FSTH_plus 512 line, the 2-rank gloo functional run of the N > 1 path.
O=$GRAFT_REPO_ROOT/gpurun_out/r03a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=r03
timeout 1500 python bench.py --shapes $O/${R}_conv_shapes_metatrain_f16.csv > $O/${R}_bench_f16.json 2> $O/${R}_bench_f16.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_meta -o ${R} -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/${R}_prof_meta.log 2>&1
python scripts/step_breakdown.py $O/${R}_prof_meta/${R}_kernel_trace.csv > $O/${R}_step_breakdown_metatrain_f16.csv 2>> $O/${R}_prof_meta.log
cp $O/${R}_prof_meta/${R}_kernel_stats.csv $O/${R}_metatrain_step_f16_kernel_stats.csv 2>/dev/null
rm -f $O/${R}_prof_meta/${R}_kernel_trace.csv
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_ft -o ${R} -- python bench.py --workload finetune_step --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/${R}_prof_ft.log 2>&1
python scripts/step_breakdown.py $O/${R}_prof_ft/${R}_kernel_trace.csv > $O/${R}_step_breakdown_finetune_f16.csv 2>> $O/${R}_prof_ft.log
cp $O/${R}_prof_ft/${R}_kernel_stats.csv $O/${R}_finetune_step_f16_kernel_stats.csv 2>/dev/null
rm -f $O/${R}_prof_ft/${R}_kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES"; do
  tag=$(echo $c | tr ' ' '+')
  timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "conv_dma_kernel" --output-format csv -d $O/${R}_pmc_gen_$tag -o ${R} -- python bench.py --workload generator --steps 2 --warmup 1 --no-cpu-baseline --no-drive > $O/${R}_pmc_gen_$tag.log 2>&1
  echo "pmc $tag rc=$?" >> $O/summary.txt
  rm -f $O/${R}_pmc_gen_$tag/${R}_kernel_trace.csv
done
python scripts/pmc_summary.py --json conv_dma_kernel $O/${R}_pmc_gen_*/*counter_collection.csv > $O/${R}_pmc_conv_dma_step.json 2> $O/${R}_pmc_summary.err
python scripts/pmc_summary.py $O/${R}_pmc_gen_*/*counter_collection.csv > $O/${R}_pmc_generator_step_f16.csv 2>> $O/${R}_pmc_summary.err
rm -rf $O/${R}_pmc_gen_*/
timeout 300 python bench.py --workload generator --generator FSTH_plus --image_size 512 --batch 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/${R}_bench_fsthplus512_f16.json 2> $O/${R}_bench_fsthplus512_f16.err
timeout 300 python bench.py --workload generator --steps 20 --warmup 5 --no-cpu-baseline > $O/${R}_bench_generator_f16.json 2> $O/${R}_bench_generator_f16.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --backend gloo > $O/${R}_bench_dp2_gloo_one_gpu_functional.json 2> $O/${R}_bench_dp2.err
PREC=2 timeout 120 python scripts/conv_micro.py > $O/${R}_conv_micro_f16.txt 2>&1
timeout 200 python scripts/bn_micro.py > $O/${R}_bn_micro_f16.txt 2>&1
LP_OVERLAP=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/${R}_bench_f16_one_stream.json 2> $O/${R}_bench_f16_one_stream.err
timeout 900 python -m pytest tests -m gpu -q -x > $O/${R}_pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?" >> $O/summary.txt
tail -3 $O/${R}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${R}_smoke.log 2>&1
echo "smoke rc=$?" >> $O/summary.txt
cut -c1-2500 $O/${R}_bench_f16.json
tail -3 $O/${R}_bench_f16.err
cat $O/${R}_pmc_conv_dma_step.json
cat $O/summary.txt
head -12 $O/${R}_step_breakdown_finetune_f16.csv
cut -c1-400 $O/${R}_bench_dp2_gloo_one_gpu_functional.json; tail -2 $O/${R}_bench_dp2.err
