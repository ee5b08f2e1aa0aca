#!/bin/bash
# round 5, call 5: replica diagnosis with buffers + embedding delta statistics (one-stream eager steps after the replays), generator parity after
# the split-K revert, default bench
O=$GRAFT_REPO_ROOT/gpurun_out/r05e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29661 scripts/dp_replica_diag.py graph 1 128 > $O/diag_graph_eager.log 2>&1
echo "== diag graph + eager rc=$?" | tee -a $O/summary.txt; grep "\[replicas\]" $O/diag_graph_eager.log | cut -c1-900 | tee -a $O/summary.txt
LP_OVERLAP=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29662 scripts/dp_replica_diag.py eager 2 128 > $O/diag_eager_onestream.log 2>&1
echo "== diag eager one-stream from the start rc=$?" | tee -a $O/summary.txt; grep "\[replicas\]" $O/diag_eager_onestream.log | cut -c1-900 | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_full_size_parity.py tests/test_generator_module.py -m gpu -q -s -p no:cacheprovider -k "generator" > $O/tests.log 2>&1; echo "gen tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|parity-256\] prec=2" $O/tests.log | cut -c1-500 | tee -a $O/summary.txt
for v in "LP_NONE=1" "LP_NONE=2"; do
  tag=$(echo $v | tr '=' '_')
  env $v timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-also --no-drive > $O/bench_$tag.json 2> $O/bench_$tag.err
  echo "metatrain $v $(python -c "import json;j=json.load(open('$O/bench_$tag.json'));print(j['ms_per_step'], j['value'], j['roofline']['frac'])" 2>&1 | tail -1)" | tee -a $O/summary.txt
done
