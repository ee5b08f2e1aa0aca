#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/c03; mkdir -p $O
cd $GRAFT_REPO_ROOT
LP_CONV_XCD=0 timeout 300 python scripts/r06/xcd_ab.py save /tmp/a.pt > $O/ab_save.txt 2>&1
for m in 2 1; do
LP_CONV_XCD=$m timeout 300 python scripts/r06/xcd_ab.py cmp /tmp/a.pt > $O/ab_cmp$m.txt 2>&1
echo "== mode $m"; grep xcd-ab $O/ab_cmp$m.txt | grep -v running | cut -c1-200; grep running $O/ab_cmp$m.txt | tail -1; tail -2 $O/ab_cmp$m.txt | cut -c1-300
done
