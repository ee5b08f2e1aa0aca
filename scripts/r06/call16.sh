#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/c16; mkdir -p $O
cd $GRAFT_REPO_ROOT
for pr in 1 2; do for ph in 0 1; do
PHASE=$ph SHAPES=ups PREC=$pr WHAT=conv timeout 200 python scripts/conv_micro.py 2>&1 | grep prec > $O/ups_p${pr}_ph$ph.txt
done; paste -d'|' $O/ups_p${pr}_ph0.txt $O/ups_p${pr}_ph1.txt | cut -c1-230; done
