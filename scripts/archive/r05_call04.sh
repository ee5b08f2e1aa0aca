#!/bin/bash
# round 5, call 4: in-kernel split-K finish + fixed deferred spectral-norm rounds: affected tests, replica diagnosis incl. eager steps after the
# replays, A/B of every new switch against the default, one-step kernel count
O=$GRAFT_REPO_ROOT/gpurun_out/r05d
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_conv_stats.py tests/test_conv_pipe.py tests/test_generator_module.py tests/test_discriminator_criterions.py tests/test_train_step.py tests/test_streams_gpu.py tests/test_prefetch.py tests/test_data_parallel_gpu.py tests/test_full_size_parity.py tests/test_fsth_plus.py tests/test_metatrain_step.py tests/test_vgg_planes.py tests/test_checkpoint_fixture.py -m gpu -q -s -p no:cacheprovider > $O/tests.log 2>&1; echo "gpu tests rc=$?" | tee $O/summary.txt
grep -E "passed|failed" $O/tests.log | tail -2 | tee -a $O/summary.txt
grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-300 | head -30 | tee -a $O/summary.txt
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29661 scripts/dp_replica_diag.py graph 2 128 > $O/diag_graph_eager.log 2>&1
echo "== diag graph + eager rc=$?" | tee -a $O/summary.txt; grep "\[replicas\]" $O/diag_graph_eager.log | cut -c1-900 | tee -a $O/summary.txt
for v in "LP_NONE=1" "LP_SPLITK_FUSED=0" "LP_SN_DEFER=0" "LP_D_POOL_RELU=0" "LP_G_Y16=0" "LP_NONE=2"; do
  tag=$(echo $v | tr '=' '_')
  env $v timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-also --no-drive > $O/bench_$tag.json 2> $O/bench_$tag.err
  echo "metatrain $v $(python -c "import json;j=json.load(open('$O/bench_$tag.json'));print(j['ms_per_step'], j['value'], j['roofline']['frac'])" 2>&1 | tail -1)" | tee -a $O/summary.txt
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_meta -o r05 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-drive > $O/prof_meta.log 2>&1
python scripts/step_breakdown.py $O/prof_meta/r05_kernel_trace.csv > $O/step_breakdown_metatrain.csv 2>> $O/prof_meta.log
rm -rf $O/prof_meta
head -3 $O/step_breakdown_metatrain.csv | cut -c1-160
