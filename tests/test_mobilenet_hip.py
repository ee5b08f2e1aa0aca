"""GPU parity of the MobileNetV2 pose encoder's HIP forward (csrc/mobilenet.hip, fp32) against fp64
torch on the CPU: the kernels one by one, then ``mobilenet_v2(256)`` whole, in eval mode (running statistics: drive.py) and in
train mode (batch statistics + running-stat update: the fine-tuning step calls the frozen embedder under no_grad in train mode).
Reference: embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:26-28,56-58 (torchvision mobilenet_v2)."""
import copy

import pytest

from oracle import backbones_ref as BR
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {0: 1e-2, 1: 3e-5, 2: 1e-3}


def _ops():
    from latent_pose_reenactment_amd import hipops
    return hipops


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize('shape', [(2, 64, 64, 32), (3, 30, 18, 8)])
def test_stem_conv_s2(shape):
    ops = _ops()
    n, h, w, cout = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, 3, h, w, generator=g, dtype=torch.float64)
    wt = torch.randn(cout, 3, 3, 3, generator=g, dtype=torch.float64) * 0.2
    ref = nhwc(F.conv2d(x, wt, None, 2, 1))
    got = ops.stem_conv_s2(x.float().cuda(), wt.float().cuda())
    assert got.shape == ref.shape
    assert rel(got, ref) < 1e-6


@pytest.mark.parametrize('affine', [False, True])
@pytest.mark.parametrize('case', [(2, 32, 32, 96, 1), (2, 32, 32, 96, 2), (3, 15, 9, 24, 2), (1, 7, 7, 960, 1), (2, 8, 8, 4, 2)])
def test_dwconv3x3(case, affine):
    ops = _ops()
    n, h, w, c, stride = case
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n, c, h, w, generator=g, dtype=torch.float64) * 3
    wt = torch.randn(c, 1, 3, 3, generator=g, dtype=torch.float64)
    sc = torch.randn(c, generator=g, dtype=torch.float64)
    sh = torch.randn(c, generator=g, dtype=torch.float64)
    a = torch.clamp(x * sc[None, :, None, None] + sh[None, :, None, None], 0, 6) if affine else x
    ref = nhwc(F.conv2d(a, wt, None, stride, 1, groups=c))
    got = ops.dwconv3x3(nhwc(x).float().cuda(), wt.float().cuda(), stride, sc.float().cuda() if affine else None,
                        sh.float().cuda() if affine else None)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert rel(got, ref) < 2e-6


@pytest.mark.parametrize('prec', [1, 0, 2])
@pytest.mark.parametrize('with_res', [False, True])
def test_affine_res_and_planes(prec, with_res):
    """x = BN(y) (+ residual) and its operand planes: the planes must reproduce a 1x1 conv of x"""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    n, h, w, c, cout = 2, 16, 16, 24, 144
    y = torch.randn(n, h, w, c, generator=g, dtype=torch.float64)
    r = torch.randn(n, h, w, c, generator=g, dtype=torch.float64)
    sc = torch.randn(c, generator=g, dtype=torch.float64)
    sh = torch.randn(c, generator=g, dtype=torch.float64)
    wt = torch.randn(cout, c, 1, 1, generator=g, dtype=torch.float64) / c ** 0.5
    ref = y * sc + sh + (r if with_res else 0)
    x, x16 = ops.affine_res(y.float().cuda(), sc.float().cuda(), sh.float().cuda(), r.float().cuda() if with_res else None, prec)
    assert rel(x, ref) < 1e-6
    assert ops.affine_res(y.float().cuda(), sc.float().cuda(), sh.float().cuda(), None).shape == y.shape
    pack = ops.pack_weights(wt.float().cuda(), 0, prec)
    got = ops.conv16(x16, pack, ksize=1, prec=prec)
    ref_c = torch.einsum('nhwc,oc->nhwo', ref, wt[:, :, 0, 0])
    assert rel(got, ref_c) < TOL[prec]


@pytest.mark.parametrize('prec', [1, 2])
def test_act_pack_relu6(prec):
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    n, h, w, c, cout = 2, 16, 16, 96, 24
    y = torch.randn(n, h, w, c, generator=g, dtype=torch.float64) * 4
    sc = torch.randn(c, generator=g, dtype=torch.float64)
    sh = torch.randn(c, generator=g, dtype=torch.float64)
    wt = torch.randn(cout, c, 1, 1, generator=g, dtype=torch.float64) / c ** 0.5
    a = ops.act_pack(y.float().cuda(), pro=3, scale=sc.float().cuda(), shift=sh.float().cuda(), prec=prec)
    got = ops.conv16(a, ops.pack_weights(wt.float().cuda(), 0, prec), ksize=1, prec=prec)
    ref = torch.einsum('nhwc,oc->nhwo', torch.clamp(y * sc + sh, 0, 6), wt[:, :, 0, 0])
    assert rel(got, ref) < TOL[prec]


def test_relu6_mean_and_running_update():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    n, h, w, c = 3, 8, 8, 1280
    y = torch.randn(n, h, w, c, generator=g, dtype=torch.float64) * 2 + 1
    bn = torch.nn.BatchNorm2d(c).double()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5); bn.bias.copy_(torch.randn(c, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(c, generator=g)); bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
    rm, rv = bn.running_mean.clone().float().cuda(), bn.running_var.clone().float().cuda()
    bn.train()
    ref = F.relu6(bn(y.permute(0, 3, 1, 2))).mean([2, 3])
    s, t = ops.bn_batch_affine(y.float().cuda(), bn.weight.float().cuda(), bn.bias.float().cuda(), rm, rv, bn.momentum, bn.eps)
    got = ops.affine_relu6_mean(y.float().cuda(), s, t)
    assert rel(got, ref) < 1e-5
    assert rel(rm, bn.running_mean) < 1e-5 and rel(rv, bn.running_var) < 1e-5


def _randomise_bn(net, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
                m.running_mean.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
                m.running_var.copy_(torch.rand(m.bias.shape, generator=g) + 0.5)
        net.classifier[1].weight.normal_(0, 0.05, generator=g)
        net.classifier[1].bias.normal_(0, 0.1, generator=g)


@pytest.mark.parametrize('case', [(2, 16, 16, 24, 144, False, False), (3, 8, 8, 96, 24, True, False), (1, 7, 5, 960, 160, True, False),
                                  (2, 32, 32, 16, 96, False, True), (8, 8, 8, 320, 1280, False, True), (4, 32, 32, 32, 192, False, True),
                                  (5, 30, 30, 96, 24, True, False)])
def test_pwconv_with_fused_batchnorm(case):
    """1x1 conv with the producer's BatchNorm (+ReLU6 / +residual) applied on load, the activated input written back, and the
    train-mode BatchNorm of the OUTPUT (scale, shift, running statistics) from the partials the same launch leaves"""
    ops = _ops()
    n, h, w, k, cout, relu6, with_res = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, h, w, k, generator=g, dtype=torch.float64) * 2
    r = torch.randn(n, h, w, k, generator=g, dtype=torch.float64)
    sc = torch.rand(k, generator=g, dtype=torch.float64) + 0.5
    sh = torch.randn(k, generator=g, dtype=torch.float64)
    wt = torch.randn(cout, k, 1, 1, generator=g, dtype=torch.float64) / k ** 0.5
    a = x * sc + sh
    if relu6:
        a = a.clamp(0, 6)
    if with_res:
        a = a + r
    ref = torch.einsum('nhwk,ok->nhwo', a, wt[:, :, 0, 0])
    bn = torch.nn.BatchNorm2d(cout).double().train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(cout, generator=g) + 0.5); bn.bias.copy_(torch.randn(cout, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(cout, generator=g)); bn.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
    rm, rv = bn.running_mean.clone().float().cuda(), bn.running_var.clone().float().cuda()
    ref_bn = bn(ref.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    y, xo, st = ops.pwconv(x.float().cuda(), wt.float().cuda(), in_scale=sc.float().cuda(), in_shift=sh.float().cuda(), in_relu6=relu6,
                           in_res=r.float().cuda() if with_res else None, want_x=True, stats=True)
    assert rel(y, ref) < 2e-6 and rel(xo, a) < 1e-6
    s, t = ops.bn_finalize(st, bn.weight.float().cuda(), bn.bias.float().cuda(), rm, rv, bn.momentum, bn.eps)
    assert rel(y * s + t, ref_bn) < 1e-5
    assert rel(rm, bn.running_mean) < 1e-5 and rel(rv, bn.running_var) < 1e-5
    y2, xo2, st2 = ops.pwconv(x.float().cuda(), wt.float().cuda())
    assert xo2 is None and st2 is None and rel(y2, torch.einsum('nhwk,ok->nhwo', x, wt[:, :, 0, 0])) < 2e-6


@pytest.mark.parametrize('case', [(2, 32, 32, 96, 2), (3, 15, 9, 24, 1), (1, 7, 7, 960, 1), (8, 16, 16, 144, 2)])
def test_dwconv3x3_with_batchnorm_statistics(case):
    ops = _ops()
    n, h, w, c, stride = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, c, h, w, generator=g, dtype=torch.float64) * 3
    wt = torch.randn(c, 1, 3, 3, generator=g, dtype=torch.float64)
    sc = torch.randn(c, generator=g, dtype=torch.float64)
    sh = torch.randn(c, generator=g, dtype=torch.float64)
    a = torch.clamp(x * sc[None, :, None, None] + sh[None, :, None, None], 0, 6)
    ref = F.conv2d(a, wt, None, stride, 1, groups=c)
    bn = torch.nn.BatchNorm2d(c).double().train()
    rm, rv = bn.running_mean.clone().float().cuda(), bn.running_var.clone().float().cuda()
    ref_bn = nhwc(bn(ref))
    y, st = ops.dwconv3x3_stats(nhwc(x).float().cuda(), wt.float().cuda(), stride, sc.float().cuda(), sh.float().cuda())
    assert rel(y, nhwc(ref)) < 2e-6
    s, t = ops.bn_finalize(st, bn.weight.float().cuda(), bn.bias.float().cuda(), rm, rv, bn.momentum, bn.eps)
    assert rel(y * s + t, ref_bn) < 1e-5
    assert rel(rm, bn.running_mean) < 1e-5 and rel(rv, bn.running_var) < 1e-5


@pytest.mark.parametrize('batch,mode', [(1, 'eval'), (4, 'eval'), (4, 'train')])
def test_mobilenet_v2_forward(batch, mode):
    """mobilenet_v2(256) on a [B, 3, 256, 256] batch: the HIP forward (taken under no_grad) vs the fp64 CPU module.  (B = 1 in train mode is
    not a case: the late 8 x 8 maps of one frame are too few positions for batch statistics to be a meaningful comparison.)"""
    from latent_pose_reenactment_amd.embedders.backbones import mobilenet_v2
    torch.manual_seed(0)
    net = mobilenet_v2(256)
    _randomise_bn(net, 7)
    net.classifier[0].p = 0.0                      # the dropout mask is random: not comparable
    ref_net = copy.deepcopy(net).double()
    dev_net = copy.deepcopy(net).cuda()
    x = torch.rand(batch, 3, 256, 256, generator=torch.Generator().manual_seed(8), dtype=torch.float64) * 2 - 1
    for m in (ref_net, dev_net):
        m.train(mode == 'train')
    with torch.no_grad():
        ref = BR.mobilenet_forward(ref_net, x)
        calls = []
        orig = dev_net._forward_hip
        dev_net._forward_hip = lambda t: (calls.append(1), orig(t))[1]
        got = dev_net(x.float().cuda())
        if mode == 'train':                        # second step: the running statistics written by the first one are inputs now
            ref = BR.mobilenet_forward(ref_net, x * 0.5)
            got = dev_net((x * 0.5).float().cuda())
    assert calls, 'the HIP forward was not taken'
    tol = 1e-4
    err = rel(got, ref)
    print(f'[parity] mobilenet_v2 {mode} B={batch}: rel-L2 {err:.3e} (tol {tol:.0e})')
    assert err < tol
    if mode == 'train':
        for (k, a), (_, b) in zip(dev_net.state_dict().items(), ref_net.state_dict().items()):
            if 'running' in k:
                assert rel(a, b) < tol, k
            if 'num_batches_tracked' in k:
                assert int(a) == int(b) == 2, k


def test_mobilenet_v2_folded_batchnorm_cache_follows_weights():
    """the folded BatchNorm (scale, shift) are cached between frames; loading a state dict must invalidate them"""
    from latent_pose_reenactment_amd.embedders.backbones import mobilenet_v2
    torch.manual_seed(1)
    a, b = mobilenet_v2(64), mobilenet_v2(64)
    _randomise_bn(a, 1); _randomise_bn(b, 2)
    x = torch.rand(2, 3, 64, 64) * 2 - 1
    dev = copy.deepcopy(a).cuda().eval()
    with torch.no_grad():
        y_a = dev(x.cuda())
        dev.load_state_dict(b.state_dict())
        y_b = dev(x.cuda())
        ref_b = BR.mobilenet_forward(b.double().eval(), x.double())
    assert rel(y_b, ref_b) < 1e-4 and rel(y_a, ref_b) > 1e-2
