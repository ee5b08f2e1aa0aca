"""XCD-ordered grids of the dense 3x3 conv kernels (csrc/conv_common.h conv16_block / conv16_grid, round 6): the order is a bijection of workgroup
ids -- which (pixel tile, Cout block) a workgroup computes, not what it computes -- so every output must be BIT-IDENTICAL with LP_CONV_XCD=0 (the
plain 2-D grid), =1 (XCD order) and =2 (1-D grid in plain order: the debug form), in both operand modes, on shapes that take conv_pipe_kernel,
the ping-pong and single-group conv_dma_kernel forms, split-K, fused upsampling and ragged tiles (scripts/r06/xcd_ab.py; the knob is read once per
process, hence the subprocesses)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_xcd_order_is_bit_identical_to_the_plain_grid(tmp_path):
    ref = str(tmp_path / 'plain.pt')
    script = os.path.join(ROOT, 'scripts', 'r06', 'xcd_ab.py')
    r = subprocess.run([sys.executable, script, 'save', ref], cwd=ROOT, env=dict(os.environ, LP_CONV_XCD='0'), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for mode in ('1', '2'):
        r = subprocess.run([sys.executable, script, 'cmp', ref], cwd=ROOT, env=dict(os.environ, LP_CONV_XCD=mode), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and '[xcd-ab] 0 mismatching cases of 32' in r.stdout, (mode, r.stdout[-3000:], r.stderr[-2000:])
