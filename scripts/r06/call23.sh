#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/c23; mkdir -p $O
cd $GRAFT_REPO_ROOT
(for pr in 1 2; do PREC=$pr timeout 200 python scripts/r06/phase_micro.py 2>&1 | grep "^prec="; done) > $O/r06_phase_conv.txt
cat $O/r06_phase_conv.txt | cut -c1-250
