#!/bin/bash
# round 6, call 5: the tightened parity tests on the new default assignment; XCD order A/B on the micro shapes (time + FETCH_SIZE); knob sweep on the step
O=$GRAFT_REPO_ROOT/gpurun_out/c05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LP_PARITY_OUT=$O timeout 1500 python -m pytest tests/test_full_size_parity.py tests/test_e1_full_gpu.py tests/test_generator_module.py -m gpu -q -s 2>&1 | grep -E "^\[parity|^\[e1|passed|failed|^FAILED|Error" | cut -c1-900 > $O/parity.txt
tail -25 $O/parity.txt | cut -c1-500
for x in 0 1; do
  LP_CONV_XCD=$x PREC=2 WHAT=conv timeout 200 python scripts/conv_micro.py > $O/micro_f16_xcd$x.txt 2>&1
  LP_CONV_XCD=$x PREC=1 WHAT=conv timeout 200 python scripts/conv_micro.py > $O/micro_x3_xcd$x.txt 2>&1
  LP_CONV_XCD=$x PREC=2 WHAT=conv REPS=4 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "conv_pipe_kernel|conv_dma_kernel" --output-format csv -d $O/pmc_f$x -o p -- python scripts/conv_micro.py > $O/pmc_f$x.log 2>&1
  python scripts/pmc_summary.py $O/pmc_f$x/*counter_collection.csv $O/pmc_f$x/*/*counter_collection.csv 2>/dev/null > $O/pmc_fetch_xcd$x.csv
  rm -rf $O/pmc_f$x
done
paste -d'|' $O/micro_f16_xcd0.txt $O/micro_f16_xcd1.txt | cut -c1-240
paste -d'|' $O/micro_x3_xcd0.txt $O/micro_x3_xcd1.txt | cut -c1-240
head -20 $O/pmc_fetch_xcd0.csv | cut -c1-200; head -20 $O/pmc_fetch_xcd1.csv | cut -c1-200
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-drive"
run() { tag=$1; shift; env "$@" timeout 300 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; python - $O/bench_$tag.json $tag <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('frac_mfma_work'))
except Exception as e: print(sys.argv[2], 'ERR',e)
P
}
run base LP_X=0
run pipex3 LP_CONV_PIPE_X3=1
run ksplit4 LP_CONV_KSPLIT=4
run splitwgs512 LP_CONV_SPLIT_WGS=512
run tail7 LP_E_F16_TAIL=7
run xcd0 LP_CONV_XCD=0
run base2 LP_X=0
for f in $O/*.err; do tail -1 $f | grep -v amdgpu.ids | cut -c1-300; done
