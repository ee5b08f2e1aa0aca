"""Discriminator + criterions: state_dict compatibility (CPU) and parity of the HIP-backed modules against golden vectors
produced by the real reference (GPU)."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def make_dis(z, padding='zero'):
    from discriminators.no_landmarks import Discriminator
    image_size, nblocks, nlabels = (int(v) for v in z['cfg'])
    return Discriminator(padding, 3, 3, 4, 16, 8, nblocks, image_size, nlabels)


PADDINGS = [('zero', 'discriminator_small.npz'), ('reflection', 'discriminator_small_reflection.npz')]      # --dis_padding (no_landmarks.py:45-50)


def test_discriminator_state_dict_matches_reference():
    z = load('discriminator_small.npz')
    D = make_dis(z)
    ref_keys = [k[3:] for k in z if k.startswith('sd.')]
    assert list(D.state_dict().keys()) == ref_keys
    assert [k for k, _ in D.named_parameters()] == [k[6:] for k in z if k.startswith('gradG.')]
    D.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('sd.')}, strict=True)
    D.enable_finetuning({'embeds': torch.zeros(1, 8)})
    assert list(D.state_dict().keys()) == [k[6:] for k in z if k.startswith('ft_sd.')]


@pytest.mark.gpu
@pytest.mark.parametrize('padding,fixture', PADDINGS)
def test_discriminator_three_passes_and_losses_vs_reference_golden(monkeypatch, padding, fixture):
    monkeypatch.setenv('LP_PREC', 'bf16x3')
    from criterions import adversarial, featmat, dice, dis_embed
    z = load(fixture)
    D = make_dis(z, padding)
    D.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('sd.')}, strict=True)
    D = D.cuda().train()
    D.keep_reference_waste = True      # this test also compares the (discarded) D-parameter gradients of loss_G
    fake = torch.from_numpy(z['fake']).cuda().requires_grad_(True)
    dd = dict(fake_rgbs=fake, target_rgbs=torch.from_numpy(z['real']).cuda(), label=torch.from_numpy(z['label']).cuda())
    D(dd)
    errs = {k: rel(dd[k], z[k]) for k in ('fake_score_G', 'fake_score_D', 'real_score', 'real_embedding')}
    for i, (f, r) in enumerate(zip(dd['fake_features'], dd['real_features'])):
        assert tuple(f.shape) == z[f'fake_feat{i}'].shape
        errs[f'fake_feat{i}'] = rel(f, z[f'fake_feat{i}']); errs[f'real_feat{i}'] = rel(r, z[f'real_feat{i}'])
    dd['fake_segm'] = torch.from_numpy(z['fake_segm']).cuda().requires_grad_(True)
    dd['real_segm'] = torch.from_numpy(z['real_segm']).cuda()
    dd['embeds_elemwise'] = torch.from_numpy(z['embeds_elemwise']).cuda().requires_grad_(True)
    lg, ld = adversarial.Criterion('gan')(dd)
    fm = featmat.Criterion(10.0)(dd)['feature_matching']
    dc = dice.Criterion(1.0)(dd)['segmentation_dice']
    de = dis_embed.Criterion(1e-2)(dd)['embedding_matching']
    for v, k in ((lg['adversarial_G'], 'loss_adv_G'), (ld['adversarial_D'], 'loss_adv_D'), (fm, 'loss_fm'), (dc, 'loss_dice'),
                 (de, 'loss_dis_embed')):
        errs[k] = rel(v, z[k])
    (lg['adversarial_G'] + fm + dc + de).backward(retain_graph=True)
    errs['gG_fake'] = rel(fake.grad, z['gG_fake']); errs['gG_fake_segm'] = rel(dd['fake_segm'].grad, z['gG_fake_segm'])
    errs['gG_elemwise'] = rel(dd['embeds_elemwise'].grad, z['gG_elemwise'])
    for k, p in D.named_parameters():
        errs['gradG.' + k] = rel(p.grad, z['gradG.' + k])
    D.zero_grad()
    ld['adversarial_D'].backward()
    for k, p in D.named_parameters():
        errs['gradD.' + k] = rel(p.grad, z['gradD.' + k])
    for k, v in D.state_dict().items():
        if k.endswith('_u') or k.endswith('_v'):
            errs['buf.' + k] = rel(v, z['sd_after.' + k])
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    print(f'[parity] discriminator + cheap criterions (bf16x3, padding={padding}): worst', [(k, f'{v:.2e}') for k, v in worst])
    # ReLU sign ties in this 4..16-channel toy net can move single gradients by a few 1e-3 (see make_golden.py note)
    bad = {k: v for k, v in errs.items() if v >= (1e-5 if k.startswith('buf.') else 5e-3 if 'grad' in k or k.startswith('gG') else 2e-4)}
    assert not bad, bad


@pytest.mark.gpu
def test_perceptual_and_vggface_vs_reference_golden(monkeypatch):
    monkeypatch.setenv('LP_PREC', 'bf16x3')
    from criterions.common.perceptual_loss import PerceptualLoss
    from criterions.idt_embed import crop_and_resize
    z = load('perceptual_small.npz')
    div = int(z['width_div'])
    fake = torch.from_numpy(z['fake']).cuda().requires_grad_(True)
    real = torch.from_numpy(z['real']).cuda()
    p19 = PerceptualLoss(3e-2, '/nonexistent', 'caffe', synthetic_seed=0, width_div=div)
    p19.model.load_state_dict({k[len('vgg19.'):]: torch.from_numpy(v) for k, v in z.items()
                               if k.startswith('vgg19.') and int(k.split('.')[1]) < 30})
    p19 = p19.cuda().eval()
    l19 = p19(fake, real)
    l19.backward()
    e1, g1 = rel(l19, z['loss_vgg19']), rel(fake.grad, z['grad_vgg19'])
    fake.grad = None
    pf = PerceptualLoss(6e-3, '/nonexistent', 'face', synthetic_seed=0, width_div=div)
    pf.model.load_state_dict({k[len('vggface.'):]: torch.from_numpy(v) for k, v in z.items()
                              if k.startswith('vggface.') and int(k.split('.')[1]) < 30})
    pf = pf.cuda().eval()
    h = w = 32
    t, l = h * (1 - 1 / 1.8) / 2, w * (1 - 1 / 1.8) / 2
    boxes = torch.tensor([[t, h - t, l, w - l]], device='cuda').expand(2, 4)
    lf = pf(crop_and_resize(fake, boxes), crop_and_resize(real, boxes))
    lf.backward()
    e2, g2 = rel(lf, z['loss_vggface']), rel(fake.grad, z['grad_vggface'])
    print(f'[parity] perceptual: loss_vgg19 {e1:.2e} grad {g1:.2e} | loss_vggface {e2:.2e} grad {g2:.2e}')
    assert e1 < 1e-4 and e2 < 1e-4 and g1 < 5e-3 and g2 < 5e-3
    assert rel(crop_and_resize(real, boxes), z['crop32']) < 1e-5


@pytest.mark.gpu
def test_spectral_norm_state_survives_many_forwards_before_backward(monkeypatch):
    """The (u, v, 1/sigma) a pass used live in rotating static buffer sets until its backward has run.  Three discriminator
    evaluations (9 passes) before ONE backward need more than the four initial sets: the ring must grow instead of overwriting a
    pass's state -- gradients equal those of evaluate-then-backward done one at a time."""
    import copy
    monkeypatch.setenv('LP_PREC', 'bf16x3')
    z = load('discriminator_small.npz')
    D0 = make_dis(z)
    D0.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('sd.')}, strict=True)
    g = torch.Generator().manual_seed(3)
    batches = [dict(fake=torch.from_numpy(z['fake']) + 0.1 * i * torch.randn(z['fake'].shape, generator=g),
                    real=torch.from_numpy(z['real']) + 0.1 * i * torch.randn(z['real'].shape, generator=g)) for i in range(3)]

    def loss_of(D, b):
        dd = dict(fake_rgbs=b['fake'].cuda(), target_rgbs=b['real'].cuda(), label=torch.from_numpy(z['label']).cuda())
        D(dd)
        return dd['fake_score_D'].square().mean() + dd['real_score'].square().mean() + sum(f.square().mean() for f in dd['real_features'])

    Da, Db = copy.deepcopy(D0).cuda().train(), copy.deepcopy(D0).cuda().train()
    sum(loss_of(Da, b) for b in batches).backward()              # A: all forwards first, one backward
    for b in batches:                                            # B: one at a time (gradients accumulate)
        loss_of(Db, b).backward()
    worst = 0.0
    for (k, pa), (_, pb) in zip(Da.named_parameters(), Db.named_parameters()):
        if pb.grad is not None and float(pb.grad.abs().max()) > 0:
            worst = max(worst, rel(pa.grad, pb.grad.cpu()))
    assert worst < 1e-4, worst
    for (k, ba), (_, bb) in zip(Da.named_buffers(), Db.named_buffers()):
        assert torch.equal(ba, bb), k                            # the power iteration itself advanced identically
