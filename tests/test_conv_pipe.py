"""The tap-pipelined 3x3 convolution kernel (csrc/conv_pipe.hip: counted-vmcnt LDS-DMA pipeline, 128 x 64 or 64 x 64 outputs per wave)
against the fp64 contraction of the same operand planes -- tests/conv_pipe_cases.py, once per wave-tile variant in a fresh interpreter
(the variant knob is read once per process), plus the default heuristics (which keep small test shapes on conv_dma_kernel: the run then
checks that both kernels give the same answers on the same cases)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize('mr', ['8', '4', 'default'])
def test_conv_pipe_kernel_vs_fp64_of_the_same_operands(mr):
    env = dict(os.environ)
    env.pop('LP_CONV_PIPE_MR', None)
    env.pop('LP_CONV1X1_PIPE', None)
    if mr != 'default':
        env['LP_CONV_PIPE_MR'] = mr
        env['LP_CONV1X1_PIPE'] = '2'           # the chunk-pipelined 1x1 kernel on every 1x1 shape
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'conv_pipe_cases.py')], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    print(r.stdout[-6000:])
    assert r.returncode == 0 and 'CONV_PIPE_OK' in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
