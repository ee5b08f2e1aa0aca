"""The VGG criterions without fp32 activations (fp16 mode): target-image taps as 16-bit planes (``_features16``) and the generated image's
pass through phantom tensors (``_features_planes``: planes-only convs, L1 on planes, plane-to-plane pools, ReLU masks from planes) against
the fp32-activation path of the same module -- loss and the gradient that reaches the generated image."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('net,size', [('caffe', 64), ('face', 96)])
def test_planes_only_vgg_matches_the_fp32_activation_path(monkeypatch, net, size):
    from latent_pose_reenactment_amd import nn as lpnn
    from latent_pose_reenactment_amd.criterions.common import perceptual_loss as pl
    if lpnn.default_prec() != lpnn.PREC_F16:
        pytest.skip('planes-only chains exist in the fp16 mode only')
    crit = pl.PerceptualLoss(0.01, '/nonexistent', net, synthetic_seed=3).cuda().eval()
    g = torch.Generator().manual_seed(1)
    fake0 = (torch.rand(2, 3, size, size, generator=g) * 2 - 1).cuda()
    real = (torch.rand(2, 3, size, size, generator=g) * 2 - 1).cuda()
    out = {}
    for name, t16, f16 in (('fp32', False, False), ('taps16', True, False), ('planes', True, True)):
        monkeypatch.setattr(pl, 'TAPS16', t16)
        monkeypatch.setattr(pl, 'FAKE16', f16)
        fake = fake0.clone().requires_grad_(True)
        loss = crit(fake, real)
        loss.backward()
        torch.cuda.synchronize()
        out[name] = (loss.detach().double().item(), fake.grad.detach().double().clone())
    l0, g0 = out['fp32']
    for name in ('taps16', 'planes'):
        l, gr = out[name]
        dl = abs(l - l0) / abs(l0)
        dg = ((gr - g0).norm() / g0.norm()).item()
        cos = (torch.dot(gr.flatten(), g0.flatten()) / (gr.norm() * g0.norm())).item()
        print(f'[vgg-planes] {net} {size}px {name}: loss {l:.6f} vs {l0:.6f} (rel {dl:.2e}); input gradient rel-L2 {dg:.2e}, cosine {cos:.6f}')
        assert dl < 1e-3, (name, dl)
        # (the L1 sign pattern flips where |a - b| is below the fp16 rounding of the features: a few 1e-4 of the sites)
        assert cos > 0.995 and dg < 0.1, (name, dg, cos)
