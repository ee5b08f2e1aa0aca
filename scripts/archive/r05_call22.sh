#!/bin/bash
# round 5, call 22: power iteration beside the LDS-DMA conv kernels of the SAME process on another stream (f16 and bf16x3 generators)
O=$GRAFT_REPO_ROOT/gpurun_out/r05v
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for p in f16 bf16x3; do
  echo "--- SN beside the generator forward (LP_PREC=$p) on a side stream"
  LP_PREC=$p SN_NOISE=conv timeout 300 python scripts/sn_determinism.py 200 2>&1 | grep -v amdgpu.ids | cut -c1-900 | tee $O/sn_beside_conv_$p.txt
done
