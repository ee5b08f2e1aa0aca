#!/bin/bash
# round 3, GPU call 9: new parity tests, conv_dma A/B (MFMA-wave priority, two ping-pong workgroups per CU), PMC over the generator step
O=$GRAFT_REPO_ROOT/gpurun_out/r03c9
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in default noprio minw4; do
  lib=""; [ $v != default ] && lib=$GRAFT_REPO_ROOT/latent_pose_reenactment_amd/liblp_hip_$v.so
  LP_LIB_OVERRIDE=$lib PREC=2 WHAT=conv REPS=30 timeout 120 python scripts/conv_micro.py > $O/conv_micro_f16_$v.txt 2>&1
done
timeout 900 python -m pytest tests/test_train_step.py tests/test_metatrain_step.py tests/test_full_size_parity.py tests/test_data_parallel_gpu.py tests/test_prefetch.py tests/test_checkpoint_fixture.py tests/test_hip_ops.py -m gpu -q -s --maxfail=80 > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/summary.txt
R=r03
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES"; do
  tag=$(echo $c | tr ' ' '+')
  timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "conv_dma_kernel" --output-format csv -d $O/${R}_pmc_gen_$tag -o ${R} -- python bench.py --workload generator --steps 2 --warmup 1 --no-cpu-baseline --no-drive > $O/${R}_pmc_gen_$tag.log 2>&1
  echo "pmc $tag rc=$?" >> $O/summary.txt
  rm -f $O/${R}_pmc_gen_$tag/${R}_kernel_trace.csv
done
python scripts/pmc_summary.py --json conv_dma_kernel $O/${R}_pmc_gen_*/*counter_collection.csv > $O/${R}_pmc_conv_dma_step.json 2> $O/${R}_pmc_summary.err
python scripts/pmc_summary.py $O/${R}_pmc_gen_*/*counter_collection.csv > $O/${R}_pmc_generator_step_f16.csv 2>> $O/${R}_pmc_summary.err
rm -rf $O/${R}_pmc_gen_*/
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-drive > $O/bench_metatrain_f16.json 2> $O/bench_metatrain_f16.err
paste $O/conv_micro_f16_default.txt $O/conv_micro_f16_noprio.txt $O/conv_micro_f16_minw4.txt | awk -F'|' '{print $1 "|" $2 "|" $4 "|" $6}' | cut -c1-200
grep -E "passed|failed|error" $O/tests.log | tail -3
grep -E "FAILED|\[parity\] (meta-train step \(Adam|train step)|\[parity-256\]|\[dp\]|input path" $O/tests.log | cut -c1-700
cat $O/${R}_pmc_conv_dma_step.json
cat $O/summary.txt
python -c "
import json; j=json.load(open('$O/bench_metatrain_f16.json')); print(j['value'], j['ms_per_step'], json.dumps(j['roofline'])[:300]); print({k: (v.get('achieved'), v.get('unit'), v.get('frac')) for k, v in j.items() if k.startswith('roofline_')})"
