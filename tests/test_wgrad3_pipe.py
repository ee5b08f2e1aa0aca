"""The DMA-staged 3x3 weight-gradient kernel (csrc/conv_wgrad.hip: wgrad3_pipe_kernel) against the fp64 contraction of the same operand
planes -- forced onto every shape (LP_WGRAD3_PIPE=2), under its default dispatch, and with it switched off (conv_wgrad_kernel on the same
cases: the A/B baseline must pass the same gate).  The address arithmetic of the kernel is also restated on the CPU
(scripts/wgrad3_pipe_emu.py, run by the non-GPU test below)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['2', '1', '0'])
def test_wgrad3_pipe_cases(mode):
    env = dict(os.environ, LP_WGRAD3_PIPE=mode)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'wgrad3_pipe_cases.py')], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-6000:])
    assert r.returncode == 0 and 'WGRAD3_PIPE_OK' in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


def test_wgrad3_pipe_address_arithmetic_on_cpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'wgrad3_pipe_emu.py')], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert 'plain 1  upsampled 1' in r.stdout, r.stdout[-500:]
