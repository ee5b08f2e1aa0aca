#!/bin/bash
# round 4, GPU call 10: where do the waves of conv_pipe_kernel spend their cycles?  SQ counters over the mid-size layer classes (micro-benchmark)
O=$GRAFT_REPO_ROOT/gpurun_out/r04c10
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --kernel-trace --kernel-include-regex "conv_pipe_kernel" --output-format csv -d $O/pmc -o r04 -- env PREC=2 WHAT=conv REPS=4 python scripts/conv_micro.py > $O/pmc.log 2>&1
echo "pmc rc=$?"
python scripts/pmc_summary.py $O/pmc/*counter_collection.csv > $O/r04_pmc_conv_pipe_micro_f16.csv 2>> $O/pmc.log
rm -rf $O/pmc
python - <<PY
import csv, collections
rows = list(csv.DictReader(open('$O/r04_pmc_conv_pipe_micro_f16.csv')))
by = collections.defaultdict(dict)
for r in rows:
    by[r['kernel']][r['counter']] = float(r['mean_per_launch_raw'])
for k, c in by.items():
    wc = c.get('SQ_WAVE_CYCLES', 0)
    if not wc: continue
    print(k[:95])
    print('   wave-cycles %.3g  wait_any %.2f  wait_inst %.2f  active_inst %.2f | mfma_busy/(gui/8*1024) %.3f | lds conflict/active %.3f' % (
        wc, c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_ANY', 0) / wc,
        c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(c.get('GRBM_GUI_ACTIVE', 1) / 8 * 1024, 1), c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
PY
