#!/bin/bash
# round 6, call 31: where do the waves of the bf16x3 1x1 conv kernel spend their cycles?  SQ counters over the identity encoder's pointwise shapes (micro-benchmark)
O=$GRAFT_REPO_ROOT/gpurun_out/r06j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --kernel-trace --kernel-include-regex "conv_dma_kernel<1" --output-format csv -d $O/pmc1 -o p -- env SHAPES=1x1 PREC=1 WHAT=conv REPS=4 python scripts/conv_micro.py > $O/pmc1.log 2>&1
echo "pmc1 rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE \
  --kernel-trace --kernel-include-regex "conv_dma_kernel<1" --output-format csv -d $O/pmc2 -o p -- env SHAPES=1x1 PREC=1 WHAT=conv REPS=4 python scripts/conv_micro.py > $O/pmc2.log 2>&1
echo "pmc2 rc=$?"; tail -3 $O/pmc2.log
for d in pmc1 pmc2; do f=$(ls $O/$d/*counter_collection.csv $O/$d/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python scripts/pmc_summary.py $f > $O/$d.csv 2>> $O/$d.log; done
rm -rf $O/pmc1 $O/pmc2
python - <<PY
import csv, collections, os
by = collections.defaultdict(dict)
for f in ('$O/pmc1.csv', '$O/pmc2.csv'):
    if not os.path.exists(f): continue
    for r in csv.DictReader(open(f)):
        by[r['kernel']][r['counter']] = float(r['mean_per_launch_raw'])
for k, c in by.items():
    print(k[:110]); print('   ', {a: '%.4g' % b for a, b in sorted(c.items())})
PY
