#!/bin/bash
# round 4, GPU call 24: which layer KINDS of the identity encoder's bf16x3 head tolerate fp16 operands (LP_E_HEAD_F16): embeds error at full geometry + time
O=$GRAFT_REPO_ROOT/gpurun_out/r04c24
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in "" "conv2" "conv1" "conv3" "conv2,conv3" "conv1,conv2" "conv1,conv3"; do
  LP_E_HEAD_F16=$k timeout 200 python scripts/e1_parity_full.py 2>&1 | grep -E "e1-parity|Error|error" | cut -c1-330 | tee -a $O/e1.txt
done
for k in "" "conv2"; do
  LP_E_HEAD_F16=$k timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-also --no-drive 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('LP_E_HEAD_F16=$k', d['ms_per_step'], 'ms')" | tee -a $O/e1.txt
done
