"""Every tile / schedule variant of the HIP kernels must pass the same per-op parity tests, not only the variant the default
heuristics pick for the (small) test shapes.  The variant knobs are read once per process (static env lookups in
csrc/conv_dma.hip, conv_wgrad.hip, conv_thin.hip; LP_THIN in hipops.py), so each setting runs tests/test_hip_ops.py in a fresh interpreter."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = [
    {'LP_CONV_PP': '1'},                        # ping-pong schedule forced on (incl. an odd tile count: masked second group)
    {'LP_CONV_PP': '0'},                        # single-group schedule everywhere
    {'LP_CONV_W8': '1'},                        # 8-wave workgroups for every shape
    {'LP_CONV_W8': '0', 'LP_CONV_KSPLIT': '1'},  # no split-K, 4-wave tiles only
    {'LP_CONV_NBUF': '3'},                      # 3-deep weight ring wherever it fits (default: only for grids of <= ~1 workgroup per CU)
    {'LP_CONV_NBUF': '2', 'LP_CONV_SPLIT_WGS': '512'},   # no ring; deeper split-K
    {'LP_WGRAD_COB': '64'},                     # 64-output-channel wgrad workgroups only
    {'LP_THIN': '0'},                           # thin-channel layers through the MFMA kernels (3- / 4-channel operand planes)
    {'LP_THIN_MFMA': '0'},                      # RGB -> 64 convs on the fp32 VALU kernel in every precision mode
]


@pytest.mark.gpu
@pytest.mark.parametrize('env', VARIANTS, ids=lambda e: ','.join(f'{k}={v}' for k, v in e.items()))
def test_hip_ops_under_variant(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_hip_ops.py'), '-q', '-m', 'gpu', '--no-header', '-x'],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f'{env}:\n{r.stdout[-3000:]}\n{r.stderr[-1500:]}'
