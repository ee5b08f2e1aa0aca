"""Kernels of one branch must give the SAME BITS whether or not another stream runs the convolution kernels beside them.

Round 5 found that they did not: the spectral-norm power iteration dropped single terms of its W^T u sums whenever it overlapped a conv of
another branch -- hipcc had compiled the accumulation to `v_pk_fma_f32 ... op_sel:[0,1,0]`, which on MI355X returns a wrong low half in
lanes 48..63 while the LDS-DMA conv kernels are co-resident (scripts/pk_forms_probe.py, profiles/r05_pk_fp32_opsel_hazard.txt,
tests/test_isa_lint.py).  Two data-parallel replicas with bit-identical weights drifted apart through it.  Here: the critic's power iteration,
the pose encoder's depthwise / affine kernels and the instance-norm statistics, alone and beside lp_conv16_fwd on a second stream."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _bits(t):
    return t.contiguous().view(torch.int32) if t.dtype == torch.float32 else t.contiguous().view(torch.int16)


@pytest.mark.parametrize('prec', ['f16', 'bf16x3'])
def test_power_iteration_beside_the_conv_kernels_is_bit_exact(prec):
    from latent_pose_reenactment_amd import hipops as ops
    from latent_pose_reenactment_amd.nn import SNBatch, SNWeight
    torch.manual_seed(0)
    p = {'f16': 2, 'bf16': 0, 'bf16x3': 1}[prec]
    x = torch.randn(8, 64, 64, 256, device='cuda')
    w = torch.randn(256, 256, 3, 3, device='cuda') * 0.02
    pk, a = ops.pack_weights(w, 0, p), ops.act_pack(x, pro=0, prec=p)
    xv = torch.randn(8, 32, 32, 512, device='cuda')
    layers = [SNWeight((512, 512, 3, 3), False, 1e-4).cuda() for _ in range(4)] + [SNWeight((13056, 512), False, 1e-4).cuda()]
    snb = SNBatch(layers)
    init = [(l.weight_u.clone(), l.weight_v.clone()) for l in layers]
    side = torch.cuda.Stream()

    def victims():
        for l, (u, v) in zip(layers, init):
            l.weight_u.copy_(u); l.weight_v.copy_(v)
        with torch.no_grad():
            st = snb.update(True)
            st = snb.update(True)
        stats = ops.instnorm_stats(xv, None, None, 1e-4)
        return {'u': torch.cat([l.weight_u for l in layers]).clone(), 'v': torch.cat([l.weight_v for l in layers]).clone(),
                'sigma': torch.stack([s[2][:2].clone() for s in st]), 'instnorm': torch.stack(stats).clone()}

    ref = victims()
    torch.cuda.synchronize()
    for i in range(24):
        if i % 4:                                        # three noisy runs, then a quiet one
            with torch.cuda.stream(side):
                for _ in range(6):
                    ops.conv16(a, pk, ksize=3, prec=p)
        cur = victims()
        torch.cuda.synchronize()
        for k in ref:
            assert torch.equal(_bits(ref[k]), _bits(cur[k])), f'run {i} ({"beside conv16" if i % 4 else "alone"}): {k} differs from the first run: max |d| {float((ref[k] - cur[k]).abs().max()):.3e}'
