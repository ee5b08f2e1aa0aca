#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/c02; mkdir -p $O
cd $GRAFT_REPO_ROOT
LP_CONV_XCD=0 timeout 300 python scripts/r06/xcd_ab.py save /tmp/a.pt > $O/ab_save.txt 2>&1
LP_CONV_XCD=1 timeout 300 python scripts/r06/xcd_ab.py cmp /tmp/a.pt > $O/ab_cmp.txt 2>&1
grep xcd-ab $O/ab_cmp.txt | cut -c1-200; tail -3 $O/ab_cmp.txt | cut -c1-300
LP_CONV_XCD=0 timeout 600 python -m pytest tests/test_full_size_parity.py -m gpu -q -x -k "discriminator and f16" 2>&1 | tail -15 | cut -c1-400 > $O/d_xcd0.txt
tail -4 $O/d_xcd0.txt
timeout 900 python -m pytest tests/test_conv_pipe.py tests/test_conv_stats.py tests/test_kernel_variants.py tests/test_resnext_hip.py -m gpu -q 2>&1 | grep -E "^FAILED|^ERROR|passed|failed" | cut -c1-300 > $O/pytest.txt
cat $O/pytest.txt | head -40
timeout 900 python -m pytest tests/test_resnext_hip.py -m gpu -q -x -k "any_batch" 2>&1 | tail -30 | cut -c1-300
