"""Pin the CPU oracle (oracle/lp_oracle.py) against fixtures produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import lp_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def T(a, grad=False):
    t = torch.from_numpy(np.array(a)).clone()
    return t.requires_grad_(True) if grad else t


def sd_from(z, prefix, grad=True):
    sd = {}
    for k, v in z.items():
        if k.startswith(prefix):
            name = k[len(prefix):]
            t = T(v)
            if grad and (name.endswith('weight_orig') or name.endswith('.bias') or name.endswith('.constant')
                         or name == 'identity_embedding'):
                t.requires_grad_(True)
            sd[name] = t
    return sd


def close(a, b, tol=2e-5, floor=1e-12):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    a = a.astype(np.float64); b = np.asarray(b, dtype=np.float64)
    err = np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), floor)      # rel-L2 (fp32 noise floor of
    assert err < tol, f'rel-L2 err {err:.3e} (tol {tol})'                               # the reference itself ~3e-6)


def test_adain_relu():
    z = load('ops_small.npz')
    y = torch.relu(O.adain(T(z['adain_x']), T(z['adain_gamma']), T(z['adain_beta'])))
    close(y, z['adain_out'])


def test_sn_power_iteration_sequence():
    z = load('ops_small.npz')
    sd = {'c.weight_orig': T(z['sn_w']), 'c.weight_u': T(z['sn_u0']), 'c.weight_v': T(z['sn_v0'])}
    for i in range(1, 4):
        w = O.sn_effective_weight(sd, 'c', O.SN_EPS_CONV, train=True)
        close(sd['c.weight_u'], z[f'sn_u{i}'])
        close(sd['c.weight_v'], z[f'sn_v{i}'])
        close(w, z[f'sn_weff{i}'])
    w = O.sn_effective_weight(sd, 'c', O.SN_EPS_CONV, train=False)
    close(w, z['sn_weff_eval'])
    close(sd['c.weight_u'], z['sn_u3'])


PADDINGS = [('zero', ''), ('reflection', '_reflection')]      # (--gen_padding / --dis_padding, fixture suffix)


@pytest.mark.parametrize('padding', ['zero', 'reflection'])
@pytest.mark.parametrize('tag,up', [('rb_ada', False), ('rb_up', True)])
def test_resblock_ada(tag, up, padding):
    z = load('ops_small.npz' if padding == 'zero' else 'reflection_ops_small.npz')
    sd = sd_from(z, tag + '.')
    sd = {k: v for k, v in sd.items() if '.' in k and not k.startswith(('grad', 'after'))}
    x = T(z[f'{tag}.x'], True)
    g0, b0, g1, b1 = (T(z[f'{tag}.{n}'], True) for n in ('g0', 'b0', 'g1', 'b1'))
    y = O.resblock_ada(x, {f'blk.{k}': v for k, v in sd.items()}, 'blk', (g0, b0), (g1, b1), up, train=True, padding=padding)
    close(y, z[f'{tag}.y'])
    y.backward(T(z[f'{tag}.gy']))
    close(x.grad, z[f'{tag}.gx'])
    for n, t in (('g0', g0), ('b0', b0), ('g1', g1), ('b1', b1)):
        close(t.grad, z[f'{tag}.grad_{n}'], 5e-5)
    for k, v in sd.items():
        if k.endswith('weight_orig') or k.endswith('bias'):
            close(v.grad, z[f'{tag}.grad.{k}'], 5e-5)
        if k.endswith('_u') or k.endswith('_v'):
            close(v, z[f'{tag}.after.{k}'])


@pytest.mark.parametrize('padding', ['zero', 'reflection'])
@pytest.mark.parametrize('tag,down', [('rb_down', True), ('rb_none', False)])
def test_resblock_none_inplace_relu_aliasing(tag, down, padding):
    z = load('ops_small.npz' if padding == 'zero' else 'reflection_ops_small.npz')
    sd = sd_from(z, tag + '.')
    sd = {f'blk.{k}': v for k, v in sd.items() if '.' in k and not k.startswith(('grad', 'after'))}
    x = T(z[f'{tag}.x'], True)
    y = O.resblock_none(torch.relu(x), sd, 'blk', down, train=True, padding=padding)
    close(y, z[f'{tag}.y'])
    close(torch.relu(x), z[f'{tag}.x_after'])       # the reference mutated its input to relu(x)
    y.backward(T(z[f'{tag}.gy']))
    close(x.grad, z[f'{tag}.gx'])
    for k, v in sd.items():
        kk = k[len('blk.'):]
        if kk.endswith('weight_orig') or kk.endswith('bias'):
            close(v.grad, z[f'{tag}.grad.{kk}'], 5e-5)


def _gen_cfg(z):
    image_size, nc, mx, e, p = (int(v) for v in z['cfg'])
    return dict(num_channels=nc, max_num_channels=mx, image_size=image_size)


def test_generator_affine_slice_order():
    z = load('generator_small.npz')
    cfg = _gen_cfg(z)
    blocks = O.generator_channels(cfg['num_channels'], cfg['max_num_channels'], cfg['image_size'])
    n_aff = sum(2 * (a + b) for a, b, _ in blocks) + 2 * blocks[-1][1]
    affs = O.split_affine_params(torch.arange(n_aff, dtype=torch.float32)[None], blocks)
    np.testing.assert_array_equal([float(b[0, 0]) for g, b in affs], z['affine_first_bias'])
    np.testing.assert_array_equal([float(g[0, 0]) for g, b in affs], z['affine_first_weight'])


@pytest.mark.parametrize('padding,suffix', PADDINGS)
def test_generator_eval_forward(padding, suffix):
    z = load(f'generator_small{suffix}.npz')
    sd = sd_from(z, 'sd.', grad=False)
    rgb, segm = O.generator_forward(sd, T(z['embeds']), T(z['pose']), train=False, padding=padding, **_gen_cfg(z))
    close(rgb, z['eval_fake_rgbs'])
    close(segm, z['eval_fake_segm'])


@pytest.mark.parametrize('padding,suffix', PADDINGS)
def test_generator_train_forward_backward(padding, suffix):
    z = load(f'generator_small{suffix}.npz')
    sd = sd_from(z, 'sd.')
    e, p = T(z['embeds'], True), T(z['pose'], True)
    rgb, segm = O.generator_forward(sd, e, p, train=True, padding=padding, **_gen_cfg(z))
    close(rgb, z['train_fake_rgbs'])
    close(segm, z['train_fake_segm'])
    ((rgb * T(z['r1'])).sum() + (segm * T(z['r2'])).sum()).backward()
    close(e.grad, z['grad_embeds'], 1e-4)
    close(p.grad, z['grad_pose'], 1e-4)
    n = 0
    for k, v in sd.items():
        if v.requires_grad:
            # skip-conv biases feed an InstanceNorm: their true gradient is 0, the reference holds ~1e-7 noise
            if k.endswith('skip.1.bias'):
                assert v.grad.abs().max() < 1e-5 and np.abs(z[f'grad.{k}']).max() < 1e-5
            else:
                close(v.grad, z[f'grad.{k}'], 1e-4)
            n += 1
        if k.endswith('_u') or k.endswith('_v'):
            close(v, z[f'sd_after.{k}'])
    assert n == sum(1 for k in z if k.startswith('grad.'))


@pytest.mark.parametrize('padding,suffix', PADDINGS)
def test_generator_finetuning_mode(padding, suffix):
    z = load(f'generator_small{suffix}.npz')
    sd = sd_from(z, 'sd.')
    for k in list(sd):
        if k.endswith('_u') or k.endswith('_v'):
            sd[k] = T(z[f'sd_after.{k}'])
    ident = T(z['ft_identity'], True)
    p = T(z['pose'], True)
    rgb, segm = O.generator_forward(sd, ident, p, train=True, padding=padding, **_gen_cfg(z))
    close(rgb, z['ft_fake_rgbs'])
    ((rgb * T(z['r1'])).sum() + (segm * T(z['r2'])).sum()).backward()
    close(ident.grad, z['ft_grad_identity'], 1e-4)
    close(p.grad, z['ft_grad_pose'], 1e-4)


@pytest.mark.parametrize('padding,suffix', PADDINGS)
def test_discriminator_and_cheap_criterions(padding, suffix):
    z = load(f'discriminator_small{suffix}.npz')
    image_size, nblocks, _ = (int(v) for v in z['cfg'])
    sd = sd_from(z, 'sd.')
    fake = T(z['fake'], True)
    real = T(z['real'])[:, 0]
    out = O.discriminator_forward(sd, fake, real, T(z['label']), image_size=image_size, dis_num_blocks=nblocks, train=True, padding=padding)
    for k in ('fake_score_G', 'fake_score_D', 'real_score', 'real_embedding'):
        close(out[k], z[k])
    assert len(out['fake_features']) == nblocks
    for i, (f, r) in enumerate(zip(out['fake_features'], out['real_features'])):
        close(f, z[f'fake_feat{i}'])
        close(r, z[f'real_feat{i}'])
    fake_segm, ee = T(z['fake_segm'], True), T(z['embeds_elemwise'], True)
    lg, ld = O.adversarial_gan(out['fake_score_G'], out['fake_score_D'], out['real_score'])
    fm = O.feature_matching(out['fake_features'], out['real_features'])
    dc = O.dice(fake_segm, T(z['real_segm']))
    de = O.dis_embed(ee, out['real_embedding'])
    for v, k in ((lg, 'loss_adv_G'), (ld, 'loss_adv_D'), (fm, 'loss_fm'), (dc, 'loss_dice'), (de, 'loss_dis_embed')):
        close(v, z[k])
    (lg + fm + dc + de).backward(retain_graph=True)
    close(fake.grad, z['gG_fake'], 1e-4)
    close(fake_segm.grad, z['gG_fake_segm'], 1e-4)
    close(ee.grad, z['gG_elemwise'], 1e-4)
    params = [k for k, v in sd.items() if v.requires_grad]
    for k in params:
        close(sd[k].grad, z[f'gradG.{k}'], 1e-4)
        sd[k].grad = None
    ld.backward()
    for k in params:
        close(sd[k].grad, z[f'gradD.{k}'], 1e-4)
    for k, v in sd.items():
        if k.endswith('_u') or k.endswith('_v'):
            close(v, z[f'sd_after.{k}'])


@pytest.mark.parametrize('padding,suffix', PADDINGS)
def test_discriminator_finetuning_embedding(padding, suffix):
    z = load(f'discriminator_small{suffix}.npz')
    image_size, nblocks, _ = (int(v) for v in z['cfg'])
    sd = sd_from(z, 'ft_sd.', grad=False)
    out = O.discriminator_forward(sd, T(z['fake']), T(z['real'])[:, 0], torch.zeros(2, dtype=torch.long),
                                  image_size=image_size, dis_num_blocks=nblocks, train=True,
                                  embed_eps=O.SN_EPS_DEFAULT, padding=padding)
    close(out['fake_score_G'], z['ft_fake_score_G'])
    close(out['real_score'], z['ft_real_score'])
    close(out['real_embedding'], z['ft_real_embedding'])


def test_perceptual_and_crop():
    z = load('perceptual_small.npz')
    div = int(z['width_div'])
    cfg19 = [v if v == 'M' else v // div for v in O.VGG19_CFG]
    cfg16 = [v if v == 'M' else v // div for v in O.VGG16_CFG]
    fake, real = T(z['fake'], True), T(z['real'])
    sd19 = {k[len('vgg19.'):]: T(v) for k, v in z.items() if k.startswith('vgg19.')}
    l19 = O.perceptual_loss(sd19, fake, real, 3e-2, cfg19)
    close(l19, z['loss_vgg19'])
    l19.backward()
    close(fake.grad, z['grad_vgg19'], 1e-4)
    fake.grad = None
    sdf = {k[len('vggface.'):]: T(v) for k, v in z.items() if k.startswith('vggface.')}
    lf = O.perceptual_loss(sdf, O.crop_and_resize_fixed(fake), O.crop_and_resize_fixed(real), 6e-3, cfg16)
    close(lf, z['loss_vggface'])
    lf.backward()
    close(fake.grad, z['grad_vggface'], 1e-4)
    close(O.crop_and_resize_fixed(real), z['crop32'])
    close(O.crop_and_resize_fixed(T(z['crop48_in'])), z['crop48'])
