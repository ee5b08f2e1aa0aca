#!/bin/bash
# NOTE: the second pass (FETCH_SIZE WRITE_SIZE) aborted inside rocprofv3 (signal 6) and ran into its timeout; no output of this call was kept.
# round 4, GPU call 31: counters of wgrad3_pipe_kernel on the layer classes of scripts/conv_micro.py SHAPES=wgrad (what bounds the kernel itself)
O=$GRAFT_REPO_ROOT/gpurun_out/r04c31
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --kernel-trace --kernel-include-regex "wgrad3_pipe_kernel" --output-format csv -d $O/pmc1 -o r04 -- env SHAPES=wgrad BIAS=1 PREC=2 WHAT=wgrad REPS=4 python scripts/conv_micro.py > $O/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE \
  --kernel-trace --kernel-include-regex "wgrad3_pipe_kernel" --output-format csv -d $O/pmc2 -o r04 -- env SHAPES=wgrad BIAS=1 PREC=2 WHAT=wgrad REPS=4 python scripts/conv_micro.py > $O/pmc2.log 2>&1
python scripts/pmc_summary.py $O/pmc1/*counter_collection.csv $O/pmc2/*counter_collection.csv > $O/r04_pmc_wgrad3_pipe_micro_f16.csv 2>> $O/pmc.log
rm -rf $O/pmc1 $O/pmc2
python - <<PY
import csv, collections
rows = list(csv.DictReader(open('$O/r04_pmc_wgrad3_pipe_micro_f16.csv')))
by = collections.defaultdict(dict)
for r in rows:
    by[r['kernel']][r['counter']] = float(r['mean_per_launch_raw'])
for k, c in by.items():
    wc = c.get('SQ_WAVE_CYCLES', 0)
    if not wc: continue
    print(k[:100])
    print('   wait_any %.2f  wait_inst %.2f  active_inst %.2f | mfma_busy %.3f | lds conflict/active %.3f | HBM MB %.1f' % (
        c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_ANY', 0) / wc,
        c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(c.get('GRBM_GUI_ACTIVE', 1) * 128, 1), c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1),
        (2 * c.get('FETCH_SIZE', 0) + c.get('WRITE_SIZE', 0)) * 1024 / 1e6))
PY
