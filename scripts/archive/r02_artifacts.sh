#!/bin/bash
# Regenerates the judged artifacts of round 2 on the GPU box (outputs under gpurun_out/r02a/, copied to profiles/ afterwards):
# bench lines (default f16 with cpu_baseline + strict-mode side line; meta-training; FSTH_plus 512), rocprofv3 kernel stats of the
# default bench command, one-step kernel breakdowns, per-shape conv tables, and the PMC passes (HBM traffic, MFMA utilisation) of
# the dominant kernel -- on the generator-only step and on the per-layer micro-benchmark (separate --pmc passes, --kernel-trace
# only, as MI355X_MICROARCH.md prescribes).
O=$GRAFT_REPO_ROOT/gpurun_out/r02a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=r02
timeout 900 python bench.py --shapes $O/${R}_conv_shapes_f16.csv > $O/${R}_bench_f16.json 2> $O/${R}_bench_f16.err
timeout 300 python bench.py --prec bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --shapes $O/${R}_conv_shapes_bf16x3.csv > $O/${R}_bench_bf16x3.json 2> $O/${R}_bench_bf16x3.err
timeout 600 python bench.py --workload metatrain_step --steps 20 --warmup 5 --no-cpu-baseline > $O/${R}_bench_metatrain_f16.json 2> $O/${R}_bench_metatrain_f16.err
timeout 300 python bench.py --workload generator --generator FSTH_plus --image_size 512 --batch 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/${R}_bench_fsthplus512_f16.json 2> $O/${R}_bench_fsthplus512_f16.err
timeout 300 python bench.py --workload generator --steps 20 --warmup 5 --no-cpu-baseline > $O/${R}_bench_generator_f16.json 2> $O/${R}_bench_generator_f16.err
for P in f16 bf16x3; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_$P -o ${R} -- python bench.py --prec $P --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/${R}_prof_$P.log 2>&1
  python scripts/step_breakdown.py $O/${R}_prof_$P/${R}_kernel_trace.csv > $O/${R}_step_breakdown_$P.csv 2>> $O/${R}_prof_$P.log
  rm -f $O/${R}_prof_$P/${R}_kernel_trace.csv
done
# PMC: HBM traffic (FETCH_SIZE, WRITE_SIZE) and MFMA utilisation of the conv kernels; each counter set in its own pass
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES"; do
  tag=$(echo $c | tr ' ' '+')
  case "$c" in *_SIZE) ;; *)       # (FETCH_SIZE / WRITE_SIZE passes over a whole step hang rocprofv3 on this pool: micro-benchmark only)
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${R}_pmc_gen_$tag -o ${R} -- python bench.py --workload generator --steps 2 --warmup 1 --no-cpu-baseline > $O/${R}_pmc_gen_$tag.log 2>&1
    rm -f $O/${R}_pmc_gen_$tag/${R}_kernel_trace.csv ;;
  esac
  PREC=2 REPS=2 WHAT=conv timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${R}_pmc_micro_$tag -o ${R} -- python scripts/conv_micro.py > $O/${R}_pmc_micro_$tag.log 2>&1
  rm -f $O/${R}_pmc_micro_$tag/${R}_kernel_trace.csv
done
python scripts/pmc_summary.py $O/${R}_pmc_gen_*/*counter_collection.csv > $O/${R}_pmc_generator_step_f16.csv 2> $O/${R}_pmc_summary.err
python scripts/pmc_summary.py $O/${R}_pmc_micro_*/*counter_collection.csv > $O/${R}_pmc_conv_micro_f16.csv 2>> $O/${R}_pmc_summary.err
python scripts/pmc_summary.py --json conv_dma_kernel $O/${R}_pmc_micro_*/*counter_collection.csv > $O/${R}_pmc_conv_dma.json 2>> $O/${R}_pmc_summary.err
python scripts/pmc_summary.py --json conv_dma_kernel $O/${R}_pmc_gen_*/*counter_collection.csv > $O/${R}_pmc_conv_dma_generator_step_mfma.json 2>> $O/${R}_pmc_summary.err
timeout 120 python scripts/mobilenet_micro.py 1 eval > $O/${R}_mobilenet_micro.txt 2>&1
timeout 120 python scripts/mobilenet_micro.py 8 train >> $O/${R}_mobilenet_micro.txt 2>&1
# functional run of the N > 1 bench path (two gloo ranks sharing the box's single GPU: NOT a performance number)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --backend gloo > $O/${R}_bench_dp2_gloo_one_gpu_functional.json 2> $O/${R}_bench_dp2.err
# CPU baseline exactly as BASELINE.md section 3 (>= 3 warm-up + >= 10 timed steps, all-core and 1-thread rows)
timeout 900 python bench.py --steps 20 --warmup 5 --no-also --cpu-baseline-full > $O/${R}_bench_cpu_baseline_full.json 2> $O/${R}_bench_cpu_baseline_full.err
PREC=2 timeout 120 python scripts/conv_micro.py > $O/${R}_conv_micro_f16.txt 2>&1
PREC=1 timeout 120 python scripts/conv_micro.py > $O/${R}_conv_micro_bf16x3.txt 2>&1
cat $O/${R}_bench_f16.json | cut -c1-1500
head -30 $O/${R}_step_breakdown_f16.csv
cat $O/${R}_pmc_conv_dma.json
