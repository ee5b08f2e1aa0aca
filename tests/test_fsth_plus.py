"""generators/FSTH_plus.py (BASELINE configs[4]): plugin with plain-Linear + LeakyReLU(0.05) projector and keypoint pose vector.
CPU: the oracle restatement against the reference-generated fixture (tests/golden/fsth_plus_small.npz).  GPU: the HIP module against
the same fixture (outputs, every parameter gradient), and the full 512 x 512, B = 4 configuration: outputs against the oracle
(B = 1 slice, both 16-bit modes) plus size-independent properties (compositing identity, batch consistency)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def load():
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'fsth_plus_small.npz'))
    return {k: z[k] for k in z.files}


def test_oracle_fsth_plus_vs_reference_golden():
    from oracle import lp_oracle as O
    z = load()
    size, nc, mx, e, p = [int(v) for v in z['cfg']]
    sd = {k[3:]: torch.from_numpy(v).clone() for k, v in z.items() if k.startswith('sd.')}
    for k, v in sd.items():
        if v.dtype.is_floating_point and not (k.endswith('_u') or k.endswith('_v')):
            v.requires_grad_(True)
    emb = torch.from_numpy(z['embeds']).requires_grad_(True)
    kp = torch.from_numpy(z['dec_keypoints']).requires_grad_(True)
    rgb, segm = O.generator_forward(sd, emb, kp[:, 0] - 0.5, num_channels=nc, max_num_channels=mx, image_size=size, train=True, fsth_plus=True)
    ((rgb * torch.from_numpy(z['r1'])).sum() + (segm * torch.from_numpy(z['r2'])).sum()).backward()
    assert rel(rgb, z['train_fake_rgbs']) < 1e-5 and rel(segm, z['train_fake_segm']) < 1e-5
    assert rel(emb.grad, z['grad_embeds']) < 1e-4 and rel(kp.grad, z['grad_kp']) < 1e-4
    for k, v in z.items():
        if k.startswith('grad.') and sd[k[5:]].grad is not None and np.abs(v).max() > 1e-4:      # (biases in front of an InstanceNorm have a zero true gradient: rounding noise)
            assert rel(sd[k[5:]].grad, v) < 2e-4, k


@pytest.mark.gpu
def test_fsth_plus_module_vs_reference_golden():
    from generators.FSTH_plus import Generator
    z = load()
    size, nc, mx, e, p = [int(v) for v in z['cfg']]
    G = Generator('zero', 3, 4, nc, mx, e, p, 'in', 4, 2, size, prec=1)
    sd_keys = list(G.state_dict())
    assert sd_keys == [k[3:] for k in z if k.startswith('sd.')], 'state_dict keys / order differ from the reference module'
    G.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('sd.')})
    G = G.cuda().train()
    emb = torch.from_numpy(z['embeds']).cuda().requires_grad_(True)
    kp = torch.from_numpy(z['dec_keypoints']).cuda().requires_grad_(True)
    dd = {'embeds': emb, 'dec_keypoints': kp}
    G(dd)
    ((dd['fake_rgbs'] * torch.from_numpy(z['r1']).cuda()).sum() + (dd['fake_segm'] * torch.from_numpy(z['r2']).cuda()).sum()).backward()
    errs = {'fake_rgbs': rel(dd['fake_rgbs'], z['train_fake_rgbs']), 'fake_segm': rel(dd['fake_segm'], z['train_fake_segm']),
            'grad_embeds': rel(emb.grad, z['grad_embeds']), 'grad_kp': rel(kp.grad, z['grad_kp'])}
    for k, prm in G.named_parameters():
        ref = z['grad.' + k]
        if np.abs(ref).max() > 1e-4:
            errs['grad.' + k] = rel(prm.grad, ref)
    for k, v in G.state_dict().items():
        if k.endswith('_u') or k.endswith('_v'):
            errs['buf.' + k] = rel(v, z['sd_after.' + k])
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print('[parity] FSTH_plus generator (bf16x3) vs reference golden: worst', [(k, f'{v:.2e}') for k, v in worst])
    assert all(v < 1e-4 for v in errs.values()), worst


@pytest.mark.gpu
@pytest.mark.parametrize('prec,tol', [(1, 1e-4), (2, 1e-3)])
def test_fsth_plus_512_batch4(prec, tol):
    """configs[4]: image_size 512 (7 up-blocks, 512-channel plateau up to 64 x 64), B = 4, 68 landmarks -> pose vector of 136"""
    from generators.FSTH_plus import Generator
    from oracle import lp_oracle as O
    torch.manual_seed(3)
    G = Generator('zero', 3, 4, 64, 512, 512, 136, 'in', 4, 2, 512, prec=prec)
    with torch.no_grad():
        G.constant.constant.normal_()
    G = G.cuda().train()
    emb, kp = torch.randn(4, 512), torch.rand(4, 1, 136)
    with torch.no_grad():
        for _ in range(4):
            G({'embeds': emb.cuda(), 'dec_keypoints': kp.cuda()})
    sd = {k: v.detach().cpu().clone() for k, v in G.state_dict().items()}
    e_c, k_c = emb.cuda().requires_grad_(True), kp.cuda().requires_grad_(True)
    dd = {'embeds': e_c, 'dec_keypoints': k_c}
    G(dd)
    assert dd['fake_rgbs'].shape == (4, 3, 512, 512) and dd['fake_segm'].shape == (4, 1, 512, 512)
    (dd['fake_rgbs'].mean() + dd['fake_segm'].mean()).backward()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p.grad).all() for p in G.parameters())
    # properties: the mask is a probability, the composite never exceeds rgb's range, samples are independent (InstanceNorm only)
    segm, rgbs = dd['fake_segm'], dd['fake_rgbs']
    assert (segm >= 0).all() and (segm <= 1).all() and (rgbs >= -0.25 * segm - 1e-6).all() and (rgbs <= 1.25 * segm + 1e-6).all()
    with torch.no_grad():
        G.eval()
        all4 = {'embeds': emb.cuda(), 'dec_keypoints': kp.cuda()}; G(all4)
        one = {'embeds': emb[2:3].cuda(), 'dec_keypoints': kp[2:3].cuda()}; G(one)
        G.train()
    # (fp16 operands: a 1e-7 difference in summation order -- other tile shapes at another batch size -- flips single fp16 roundings)
    assert rel(all4['fake_rgbs'][2:3], one['fake_rgbs'].cpu()) < (1e-5 if prec == 1 else 1e-3)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        rgb, sg = O.generator_forward(sd, emb[:1], kp[:1, 0] - 0.5, num_channels=64, max_num_channels=512, image_size=512, train=True, fsth_plus=True)
    e1, e2 = rel(rgbs[:1], rgb), rel(segm[:1], sg)
    print(f'[parity-512] FSTH_plus 512x512 B=4 prec={prec}: fake_rgbs {e1:.2e} fake_segm {e2:.2e} vs oracle')
    assert e1 < tol and e2 < tol
