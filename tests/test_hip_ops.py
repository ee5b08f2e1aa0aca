"""GPU parity tests of the individual HIP kernels (through the C ABI) against fp64 torch-CPU references.
Tolerances (rel-L2): bf16x3 ('exact') mode 3e-5 -- fp32-class; bf16 mode 1e-2 -- operand rounding 2^-9; f16 mode 1e-3 -- operand
rounding 2^-12 (gradient operands additionally scaled by a power of two from their amax)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {0: 1e-2, 1: 3e-5, 2: 1e-3}
TOL_DB = {0: 3e-3, 1: 1e-5, 2: 5e-4}      # bias gradient = column sums of the packed dy planes


def _ops():
    from latent_pose_reenactment_amd import hipops
    return hipops


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def report(name, err, tol):
    print(f'[parity] {name}: rel-L2 {err:.3e} (tol {tol:.0e})')
    assert err < tol, f'{name}: rel-L2 {err:.3e} >= {tol}'


def act_ref(x, pro, scale, shift, ups):
    a = x
    if pro == 1:
        a = torch.relu(x * scale[:, :, None, None] + shift[:, :, None, None])
    elif pro == 2:
        a = torch.relu(x)
    if ups:
        a = a.repeat_interleave(2, 2).repeat_interleave(2, 3)
    return a


CONV_CASES = [
    # N, H(out), W, Cin, Cout, ks, ups, pro, bias, res_shift(-1 none)
    (2, 8, 8, 64, 128, 3, 0, 1, 0, -1),
    (8, 4, 4, 64, 64, 3, 0, 1, 0, 0),
    (3, 4, 4, 128, 256, 3, 0, 1, 0, -1),
    (1, 16, 16, 128, 64, 3, 1, 1, 0, -1),
    (2, 8, 8, 64, 192, 3, 1, 1, 0, 1),
    (2, 32, 32, 64, 64, 3, 1, 1, 0, 1),
    (2, 32, 32, 64, 4, 3, 0, 1, 1, -1),
    (2, 16, 16, 4, 64, 3, 0, 0, 0, -1),
    (2, 16, 16, 3, 64, 3, 0, 0, 1, -1),
    (2, 8, 8, 64, 128, 1, 0, 0, 1, -1),
    (2, 16, 16, 128, 24, 1, 0, 2, 1, -1),
    (1, 32, 32, 16, 8, 3, 0, 2, 1, 0),
    (2, 64, 64, 64, 64, 3, 0, 1, 0, -1),
    (2, 20, 12, 3, 128, 1, 0, 0, 1, -1),      # direct fp32 kernel for <= 4 input channels: 1x1, odd image size
    (3, 12, 20, 3, 64, 3, 0, 0, 1, -1),
    (3, 8, 16, 64, 128, 3, 0, 1, 1, -1),        # 3 tiles: with the ping-pong schedule the last workgroup's second group is masked
    (5, 16, 16, 64, 64, 3, 0, 2, 1, 0),         # 5 tiles of 256 pixels (Cout <= 64 tile), residual
]


@pytest.mark.parametrize('prec', [1, 0, 2])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd(case, prec):
    ops = _ops()
    n, h, w, cin, cout, ks, ups, pro, has_bias, rs = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    hin, win = (h // 2, w // 2) if ups else (h, w)
    x = torch.randn(n, cin, hin, win, generator=g, dtype=torch.float64)
    wgt = torch.randn(cout, cin, ks, ks, generator=g, dtype=torch.float64) / (cin * ks * ks) ** 0.5
    scale = torch.randn(n, cin, generator=g, dtype=torch.float64)
    shift = torch.randn(n, cin, generator=g, dtype=torch.float64) * 0.5
    bias = torch.randn(cout, generator=g, dtype=torch.float64) if has_bias else None
    res = torch.randn(n, cout, h >> rs, w >> rs, generator=g, dtype=torch.float64) if rs >= 0 else None
    alpha = torch.tensor(0.73, dtype=torch.float64)
    ref = alpha * F.conv2d(act_ref(x, pro, scale, shift, ups), wgt, None, 1, ks // 2)
    if bias is not None:
        ref = ref + bias[None, :, None, None]
    if res is not None:
        r = res
        for _ in range(rs):
            r = r.repeat_interleave(2, 2).repeat_interleave(2, 3)
        ref = ref + r
    dev = 'cuda'
    f32 = lambda t: None if t is None else t.float().to(dev).contiguous()
    small_k = (ks == 3 and not ups and cin <= 32)
    pack = ops.pack_weights(f32(wgt), 0, prec, small_k=small_k)
    y = ops.conv(f32(nhwc(x)), pack, ksize=ks, upsample=bool(ups), pro=pro, scale=f32(scale), shift=f32(shift), bias=f32(bias),
                 res=None if res is None else f32(nhwc(res)), res_shift=max(rs, 0), alpha=f32(alpha), prec=prec)
    torch.cuda.synchronize()
    report(f'conv_fwd{case} prec={prec}', rel(y.permute(0, 3, 1, 2), ref), TOL[prec])


@pytest.mark.parametrize('prec', [1, 0, 2])
@pytest.mark.parametrize('case', [(2, 8, 8, 64, 128, 3), (2, 16, 16, 64, 4, 3), (2, 16, 16, 128, 64, 1), (8, 4, 4, 128, 64, 3),
                                  (2, 32, 32, 4, 4, 3), (2, 32, 32, 8, 4, 3), (1, 64, 64, 64, 4, 3)])
def test_conv_dgrad(case, prec):
    """data gradient = same kernel on dY with the mode-1 (flipped, transposed) pack"""
    ops = _ops()
    n, h, w, cin, cout, ks = case
    g = torch.Generator().manual_seed(5)
    wgt = torch.randn(cout, cin, ks, ks, generator=g, dtype=torch.float64) / (cin * ks * ks) ** 0.5
    dy = torch.randn(n, cout, h, w, generator=g, dtype=torch.float64)
    a = torch.zeros(n, cin, h, w, dtype=torch.float64, requires_grad=True)
    F.conv2d(a, wgt, None, 1, ks // 2).backward(dy)
    f32 = lambda t: t.float().cuda().contiguous()
    pack = ops.pack_weights(f32(wgt), 1, prec, small_k=(ks == 3 and cout <= 32))
    da = ops.conv(f32(nhwc(dy)) * 1e-6, pack, ksize=ks, prec=prec, grad=True) * 1e6      # (tiny gradients: fp16 needs its input scale)
    torch.cuda.synchronize()
    report(f'conv_dgrad{case} prec={prec}', rel(da.permute(0, 3, 1, 2), a.grad), TOL[prec])


@pytest.mark.parametrize('prec', [1, 0, 2])
@pytest.mark.parametrize('case', [(2, 16, 16, 64, 128, 3), (8, 4, 4, 128, 256, 3), (2, 32, 32, 6, 64, 3), (2, 64, 64, 64, 64, 1)])
def test_conv_dgrad_fused_relu_mask(case, prec):
    """dgrad launch with the ReLU backward fused in its epilogue == separate conv + lp_relu_bwd (coalesced, split-K and
    narrow-channel epilogues)"""
    ops = _ops()
    n, h, w, cin, cout, ks = case
    g = torch.Generator().manual_seed(9)
    wgt = (torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5).cuda()
    dy = torch.randn(n, h, w, cout, generator=g).cuda()
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    pack = ops.pack_weights(wgt, 1, prec, small_k=(ks == 3 and cout <= 32))
    two = ops.relu_bwd(ops.conv(dy, pack, ksize=ks, prec=prec), x)
    one = ops.conv(dy, pack, ksize=ks, prec=prec, relu_mask=x)
    torch.cuda.synchronize()
    assert torch.equal((one == 0), (two == 0))
    report(f'dgrad_mask{case} prec={prec}', rel(one, two), 1e-6)


WGRAD_CASES = [
    # N, H(out), W, Cin, Cout, ks, ups, pro
    (2, 8, 8, 64, 64, 3, 0, 1),
    (8, 4, 4, 64, 128, 3, 0, 1),
    (2, 16, 16, 128, 64, 3, 1, 1),
    (1, 32, 32, 64, 4, 3, 0, 1),
    (2, 16, 16, 64, 128, 1, 0, 0),
    (2, 32, 32, 16, 8, 3, 1, 1),
    (2, 16, 16, 3, 64, 3, 0, 0),
    (2, 64, 64, 64, 64, 3, 0, 1),
    (2, 20, 12, 3, 64, 1, 0, 0),        # thin-side fp32 kernels: 1x1 image-side skip conv, odd image size
    (3, 12, 20, 3, 128, 3, 0, 0),
    (2, 20, 12, 128, 4, 3, 0, 2),       # generator-head shape with a plain ReLU prologue
    (2, 16, 16, 64, 3, 3, 0, 0),
]


@pytest.mark.parametrize('prec', [1, 0, 2])
@pytest.mark.parametrize('case', WGRAD_CASES)
def test_conv_wgrad(case, prec):
    ops = _ops()
    n, h, w, cin, cout, ks, ups, pro = case
    g = torch.Generator().manual_seed(11)
    hin, win = (h // 2, w // 2) if ups else (h, w)
    x = torch.randn(n, cin, hin, win, generator=g, dtype=torch.float64)
    scale = torch.randn(n, cin, generator=g, dtype=torch.float64)
    shift = torch.randn(n, cin, generator=g, dtype=torch.float64) * 0.5
    dy = torch.randn(n, cout, h, w, generator=g, dtype=torch.float64)
    wgt = torch.zeros(cout, cin, ks, ks, dtype=torch.float64, requires_grad=True)
    F.conv2d(act_ref(x, pro, scale, shift, ups), wgt, None, 1, ks // 2).backward(dy)
    f32 = lambda t: t.float().cuda().contiguous()
    dw, db = ops.conv_wgrad(f32(nhwc(x)), f32(nhwc(dy)), ksize=ks, upsample=bool(ups), pro=pro, scale=f32(scale), shift=f32(shift), prec=prec,
                            bias_grad=True)
    torch.cuda.synchronize()
    report(f'conv_wgrad{case} prec={prec}', rel(dw, wgt.grad), TOL[prec])
    report(f'conv_wgrad bias grad{case}', rel(db, dy.sum(dim=(0, 2, 3))), TOL_DB[prec])       # column sums of dy, same launch


@pytest.mark.parametrize('shape', [(2, 4, 4, 64), (3, 16, 16, 128), (2, 128, 128, 16), (1, 32, 32, 8), (2, 32, 32, 4)])
def test_instnorm_stats(shape):
    ops = _ops()
    n, h, w, c = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, c, h, w, generator=g, dtype=torch.float64) * 3 + 10.0      # large mean: cancellation check
    aff = torch.randn(n, 3 * c, generator=g, dtype=torch.float64)
    gamma, beta = aff[:, c:2 * c], aff[:, :c]
    mean = x.mean((2, 3)); var = x.var((2, 3), unbiased=False); rstd = 1 / torch.sqrt(var + 1e-4)
    affc = aff.float().cuda()
    m, r, sc, sh = ops.instnorm_stats(nhwc(x).float().cuda(), affc[:, c:2 * c], affc[:, :c], 1e-4)
    torch.cuda.synchronize()
    report('mean', rel(m, mean), 1e-6); report('rstd', rel(r, rstd), 2e-5)
    report('scale', rel(sc, rstd * gamma), 2e-5); report('shift', rel(sh, beta - mean * rstd * gamma), 2e-4)


@pytest.mark.parametrize('prec', [0, 1])
@pytest.mark.parametrize('ups', [0, 1])
@pytest.mark.parametrize('shape', [(2, 4, 4, 64), (2, 16, 16, 32), (1, 128, 128, 8), (2, 8, 8, 16)])
def test_gradient_producers_write_operand_planes_directly(shape, ups, prec):
    """(round 6) lp_adain_relu_bwd_planes / lp_sum2x2_planes: the bf16 / bf16x3 operand planes written by the producing launch are BIT-IDENTICAL to
    lp_act_pack of the fp32 result of the two-launch form; keep_dx=False returns no fp32 tensor"""
    ops = _ops()
    n, h, w, c = shape
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(n, h, w, c, generator=g) * 2 + 1).cuda()
    aff = torch.randn(n, 2 * c, generator=g).cuda()
    dA = torch.randn(n, h << ups, w << ups, c, generator=g).cuda()
    add = torch.randn(n, h, w, c, generator=g).cuda()
    m, r, sc, sh = ops.instnorm_stats(x, aff[:, c:], aff[:, :c], 1e-4)
    d0, d1, d2 = torch.zeros_like(aff), torch.zeros_like(aff), torch.zeros_like(aff)
    ref = ops.adain_relu_bwd(dA, x, add, aff[:, c:], m, r, sc, sh, d0[:, c:], d0[:, :c], bool(ups))
    ref16 = ops.act_pack(ref, prec=prec, grad=True)
    dx, p16 = ops.adain_relu_bwd(dA, x, add, aff[:, c:], m, r, sc, sh, d1[:, c:], d1[:, :c], bool(ups), planes=prec)
    none, q16 = ops.adain_relu_bwd(dA, x, add, aff[:, c:], m, r, sc, sh, d2[:, c:], d2[:, :c], bool(ups), planes=prec, keep_dx=False)
    torch.cuda.synchronize()
    assert none is None and torch.equal(dx, ref) and torch.equal(d0, d1) and torch.equal(d0, d2)
    for got in (p16, q16):
        assert got.inv is None and torch.equal(got.hi, ref16.hi) and (prec == 0 or torch.equal(got.lo, ref16.lo)) and (got.lo is None) == (prec == 0)
    if ups:
        s_ref = ops.act_pack(ops.sum2x2(dA), prec=prec, grad=True)
        s16 = ops.sum2x2_planes(dA, prec)
        torch.cuda.synchronize()
        assert torch.equal(s16.hi, s_ref.hi) and (prec == 0 or torch.equal(s16.lo, s_ref.lo))


@pytest.mark.parametrize('with_embed', [True, False])
@pytest.mark.parametrize('shape', [(8, 4, 4, 512), (3, 2, 5, 40), (1, 1, 1, 300)])
def test_projection_head_of_the_critic(shape, with_embed):
    """(round 6) nn.ProjScoreFn = relu -> sum over (H, W) -> <pooled, embed> (no_landmarks.py:100-108) against torch autograd in fp64"""
    from latent_pose_reenactment_amd import nn as lpnn
    n, h, w, c = shape
    g = torch.Generator().manual_seed(9)
    out = torch.randn(n, h, w, c, generator=g).cuda().requires_grad_(True)
    emb = torch.randn(n, c, generator=g).cuda().requires_grad_(True) if with_embed else None
    r1, r2 = torch.randn(n, c, generator=g).cuda(), torch.randn(n, generator=g).cuda()
    pooled, dot = lpnn.ProjScoreFn.apply(out, emb)
    loss = (pooled * r1).sum() + ((dot * r2).sum() if with_embed else 0)
    loss.backward()
    o64 = out.detach().double().requires_grad_(True)
    e64 = emb.detach().double().requires_grad_(True) if with_embed else None
    p64 = torch.relu(o64).sum(dim=(1, 2))
    l64 = (p64 * r1.double()).sum() + (((p64 * e64).sum(1) * r2.double()).sum() if with_embed else 0)
    l64.backward()
    torch.cuda.synchronize()
    rel = lambda a, b: ((a.double() - b).norm() / b.norm().clamp_min(1e-30)).item()
    assert rel(pooled, p64) < 1e-6 and rel(out.grad, o64.grad) < 1e-6
    if with_embed:
        assert rel(dot, (p64 * e64).sum(1)) < 1e-5 and rel(emb.grad, e64.grad) < 1e-6


def test_vgg_input_preparation_matches_the_reference_expression_bit_for_bit():
    """(round 6) nn.ImagePrepFn: ((x + 1) / 2 - mean) / std + NCHW -> NHWC in one launch == the torch expression of perceptual_loss.py:72-93, and its
    backward == autograd of that expression"""
    from latent_pose_reenactment_amd import nn as lpnn
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(3, 3, 20, 36, generator=g) * 2 - 1).cuda().requires_grad_(True)
    mean = (torch.tensor([103.939, 116.779, 123.680]) / 255.).cuda()
    std = (torch.tensor([1., 1., 1.]) / 255.).cuda()
    y = lpnn.ImagePrepFn.apply(x, mean, std)
    r = torch.randn(y.shape, generator=g).cuda()
    (y * r).sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    y2 = (((x2 + 1) / 2) - mean[None, :, None, None]) / std[None, :, None, None]
    (y2.permute(0, 2, 3, 1) * r).sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(y, y2.permute(0, 2, 3, 1)) and torch.equal(x.grad, x2.grad)


@pytest.mark.parametrize('prec', [0, 1, 2])
@pytest.mark.parametrize('masked', [True, False])
@pytest.mark.parametrize('shape', [(2, 4, 4, 64), (3, 5, 7, 8), (1, 64, 64, 128)])
def test_pool_grad_pack_matches_the_unfused_chain(shape, masked, prec):
    """(round 6) lp_pool_grad_pack: dm = dy * [y > 0], its operand planes and the operand planes of the pool's adjoint 0.25 * up2(dm) in ONE launch
    against where + lp_act_pack + lp_avgpool2_bwd + lp_act_pack -- bit-identical planes in the bf16 modes; fp16: identical DECODED values
    (planes * 1/scale), the scale being taken from amax(dy) instead of amax(dm)"""
    ops = _ops()
    n, h, w, c = shape
    g = torch.Generator().manual_seed(5)
    dy = (torch.randn(n, h, w, c, generator=g) * 3e-3).cuda()
    y = torch.relu(torch.randn(n, h, w, c, generator=g)).cuda() if masked else None
    dm_ref = dy if y is None else torch.where(y > 0, dy, torch.zeros((), device='cuda'))
    lo_ref = ops.act_pack(dm_ref, prec=prec, grad=True)
    up_ref = ops.act_pack(ops.avgpool2_bwd(dy, None, False, y_relu=y), prec=prec, grad=True)
    dm, lo, up = ops.pool_grad_pack(dy, y, prec, want_dm=True, want_lo=True, want_up=True)
    torch.cuda.synchronize()
    assert torch.equal(dm, dm_ref)
    if prec != 2:
        assert lo.inv is None and up.inv is None
        assert torch.equal(lo.hi, lo_ref.hi) and torch.equal(up.hi, up_ref.hi)
        if prec == 1:
            assert torch.equal(lo.lo, lo_ref.lo) and torch.equal(up.lo, up_ref.lo)
    else:
        dec = lambda a: a.hi.view(torch.float16).double() * a.inv.double()
        # fp16 operands carry 11 significant bits; the two scales differ by a power of two at most (amax(dy) >= amax(dm)): same values unless a
        # tiny entry falls into the subnormal range of the coarser scale
        assert (dec(lo) - dec(lo_ref)).abs().max() <= 2.0 ** -11 * dm_ref.abs().max().double() * 2 ** -10
        assert (dec(up) - dec(up_ref)).abs().max() <= 2.0 ** -11 * dm_ref.abs().max().double() * 2 ** -10
        assert torch.equal(up.hi[:, ::2, ::2], lo.hi) and torch.equal(up.hi[:, 1::2, 1::2], lo.hi)
    none, lo2, none2 = ops.pool_grad_pack(dy, y, prec, want_dm=False, want_lo=True, want_up=False)
    torch.cuda.synchronize()
    assert none is None and none2 is None and torch.equal(lo2.hi, lo.hi)


@pytest.mark.parametrize('ups', [0, 1])
@pytest.mark.parametrize('shape', [(2, 4, 4, 64), (2, 16, 16, 32), (1, 128, 128, 8), (2, 32, 32, 4), (2, 8, 8, 16)])
def test_adain_relu_bwd(shape, ups):
    ops = _ops()
    n, h, w, c = shape
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(n, c, h, w, generator=g, dtype=torch.float64) * 2 + 1).requires_grad_(True)
    aff = torch.randn(n, 2 * c, generator=g, dtype=torch.float64).requires_grad_(True)
    gamma, beta = aff[:, c:], aff[:, :c]
    mean = x.mean((2, 3), keepdim=True); var = x.var((2, 3), unbiased=False, keepdim=True)
    a = torch.relu((x - mean) / torch.sqrt(var + 1e-4) * gamma[:, :, None, None] + beta[:, :, None, None])
    if ups:
        a = a.repeat_interleave(2, 2).repeat_interleave(2, 3)
    dA = torch.randn(a.shape, generator=g, dtype=torch.float64)
    add = torch.randn(n, c, h, w, generator=g, dtype=torch.float64)
    a.backward(dA)
    affc = aff.detach().float().cuda()
    xc = nhwc(x.detach()).float().cuda()
    m, r, sc, sh = ops.instnorm_stats(xc, affc[:, c:], affc[:, :c], 1e-4)
    daff = torch.zeros_like(affc)
    dx = ops.adain_relu_bwd(nhwc(dA).float().cuda(), xc, nhwc(add).float().cuda(), affc[:, c:], m, r, sc, sh, daff[:, c:], daff[:, :c],
                            bool(ups))
    torch.cuda.synchronize()
    report('dx', rel(dx.permute(0, 3, 1, 2), x.grad + add), 2e-5)
    report('daffine', rel(daff, aff.grad), 2e-5)


def test_sum2x2_and_head():
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 8, 16, 16, generator=g, dtype=torch.float64)
    s = ops.sum2x2(nhwc(x).float().cuda())
    report('sum2x2', rel(s.permute(0, 3, 1, 2), F.avg_pool2d(x, 2) * 4), 1e-6)
    z = torch.randn(2, 4, 16, 16, generator=g, dtype=torch.float64, requires_grad=True)
    t = torch.tanh(z)
    rgb = t[:, :3] * 0.75 + 0.5; segm = t[:, 3:] * 0.5 + 0.5
    fake = rgb * segm
    d1 = torch.randn(fake.shape, generator=g, dtype=torch.float64); d2 = torch.randn(segm.shape, generator=g, dtype=torch.float64)
    ((fake * d1).sum() + (segm * d2).sum()).backward()
    tt, r, sg = ops.head_fwd(nhwc(z.detach()).float().cuda())
    dz = ops.head_bwd(tt, d1.float().cuda(), d2.float().cuda())
    torch.cuda.synchronize()
    report('fake_rgbs', rel(r, fake), 1e-6); report('fake_segm', rel(sg, segm), 1e-6)
    report('head dz', rel(dz.permute(0, 3, 1, 2), z.grad), 1e-5)


@pytest.mark.parametrize('shape', [(8, 13056, 768), (1, 768, 768), (3, 1, 512), (11, 40, 64)])       # B, N, K (projector, drive B=1, critic head, two batch tiles)
def test_linear_fwd_bwd(shape):
    ops = _ops()
    b, n, k = shape
    g = torch.Generator().manual_seed(21)
    x = torch.randn(b, k, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(n, k, generator=g, dtype=torch.float64) / k ** 0.5).requires_grad_(True)
    bias = torch.randn(n, generator=g, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(b, n, generator=g, dtype=torch.float64)
    alpha = 0.37
    (alpha * (x @ w.t()) + bias).backward(gy)
    f32 = lambda t: t.detach().float().cuda().contiguous()
    al = torch.tensor([alpha], dtype=torch.float32, device='cuda')
    y = ops.linear_fwd(f32(x), f32(w), f32(bias), al)
    dx, dw, db = ops.linear_bwd(f32(x), f32(w), f32(gy), al, True, True, True)
    torch.cuda.synchronize()
    report(f'linear_fwd{shape}', rel(y, alpha * (x @ w.t()) + bias), 2e-6)
    report(f'linear dx{shape}', rel(dx, x.grad), 2e-6)
    report(f'linear dw (raw){shape}', rel(dw * alpha, w.grad), 2e-6)        # the kernel returns the gradient w.r.t. W/sigma
    report(f'linear db{shape}', rel(db, bias.grad), 2e-6)


@pytest.mark.parametrize('case', [(2, 3, 32, 32, 32, 32, 1 / 1.8), (1, 2, 48, 40, 48, 40, 1 / 1.8), (2, 3, 24, 24, 32, 32, 1.3)])   # last: bbox beyond the image
def test_grid_crop_fwd_bwd(case):
    """crop_and_resize (idt_embed.py:58-83) incl. reflection when the box leaves the image, against torch's own affine_grid +
    grid_sample in fp64"""
    ops = _ops()
    n, c, h, w, ho, wo, keep = case
    g = torch.Generator().manual_seed(5)
    img = torch.rand(n, c, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    t, l = h * (1 - keep) / 2 + 0.3, w * (1 - keep) / 2 - 0.2
    boxes = torch.tensor([[t, h - t, l, w - l]], dtype=torch.float64).expand(n, 4).contiguous()
    theta = torch.zeros(n, 2, 3, dtype=torch.float64)
    bt, bb, bl, br = boxes.t()
    theta[:, 0, 0] = (br - bl) / w; theta[:, 0, 2] = (bl + br) / w - 1
    theta[:, 1, 1] = (bb - bt) / h; theta[:, 1, 2] = (bt + bb) / h - 1
    grid = F.affine_grid(theta, (n, c, ho, wo), align_corners=False)
    ref = F.grid_sample(img, grid, mode='bilinear', padding_mode='reflection', align_corners=False)
    gout = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(gout)
    out = ops.grid_crop_fwd(img.detach().float().cuda(), boxes.float().cuda(), (ho, wo))
    dimg = ops.grid_crop_bwd(gout.float().cuda(), boxes.float().cuda(), (n, c, h, w))
    torch.cuda.synchronize()
    report(f'grid_crop fwd{case}', rel(out, ref), 1e-5)
    report(f'grid_crop bwd{case}', rel(dimg, img.grad), 1e-5)


def test_producer_side_amax_matches_the_amax_pass():
    """fp16 gradient operands: max|x| folded into the producing kernel's epilogue (conv data gradient incl. split-K, AdaIN backward,
    avg-pool backward, L1 backward, 2x2 sum) gives the same scale and the same planes as the separate amax pass; an in-place
    update of the tensor invalidates the recorded maximum (version counter)."""
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    prec = 2

    def packs_equal(t):
        assert getattr(t, '_lp_amax', None) is not None
        before = dict(ops.AMAX_STATS)
        a = ops.act_pack(t, prec=prec, grad=True)
        assert ops.AMAX_STATS['fused'] == before['fused'] + 1, 'recorded maximum not used'
        ops.FUSE_AMAX = False
        try:
            b = ops.act_pack(t, prec=prec, grad=True)
        finally:
            ops.FUSE_AMAX = True
        assert torch.equal(a.inv, b.inv) and torch.equal(a.hi, b.hi)

    for (n, h, w, cin, cout) in [(2, 4, 4, 512, 512), (2, 32, 32, 128, 64), (1, 16, 16, 64, 6)]:       # split-K, coalesced, element-wise epilogues
        dy = (torch.randn(n, h, w, cout, generator=g) * 1e-4).cuda()
        wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).cuda()
        d16 = ops.act_pack(dy, prec=prec, grad=True)
        dx = ops.conv16(d16, ops.pack_weights(wt, 1, prec), ksize=3, prec=prec, amax=True)
        packs_equal(dx)
    x = (torch.randn(2, 16, 16, 64, generator=g) * 3e-5).cuda()
    packs_equal(ops.avgpool2_bwd(torch.randn(2, 8, 8, 64, generator=g).cuda() * 1e-3, x, True, amax=True))
    packs_equal(ops.sum2x2(x, amax=True))
    packs_equal(ops.l1_bwd(x, x.flip(0), torch.tensor(2e-3).cuda(), 1.0 / x.numel(), True, add=x * 0.1, amax=True))
    t = ops.sum2x2(x, amax=True)
    t.mul_(64.0)                               # in-place: the recorded maximum is stale now
    before = dict(ops.AMAX_STATS)
    a = ops.act_pack(t, prec=prec, grad=True)
    assert ops.AMAX_STATS['pass'] == before['pass'] + 1
    ref = t.double().cpu()
    back = a.hi.view(torch.float16).double().cpu() * float(a.inv[0])
    assert rel(back, ref) < 1e-3


def test_recorded_amax_survives_autograd():
    """the maximum recorded by a data-gradient launch is found again by the next layer's backward (same tensor object through the
    autograd engine): a conv -> relu -> conv chain packs its inner gradient without an amax pass"""
    ops = _ops()
    from latent_pose_reenactment_amd.nn import hip_conv
    g = torch.Generator().manual_seed(32)
    x = torch.randn(2, 16, 16, 64, generator=g).cuda().requires_grad_(True)
    w1 = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).cuda().requires_grad_(True)
    w2 = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).cuda().requires_grad_(True)
    y = hip_conv(hip_conv(x, w1, ksize=3, prec=2), w2, ksize=3, pro=2, prec=2)
    before = dict(ops.AMAX_STATS)
    y.square().sum().backward()
    assert ops.AMAX_STATS['fused'] - before['fused'] >= 1, ops.AMAX_STATS


@pytest.mark.parametrize('prec', [1, 2])
def test_pack_batch_pairs_equal_single_packs(prec):
    """PackBatch packs a weight that is needed in both orientations with lp_pack_weights_pairs (one read of W through an LDS tile):
    bit-identical to lp_pack_weights, padding included, for 3x3 / 1x1 kernels, odd channel counts and the small-k padding"""
    ops = _ops()
    g = torch.Generator().manual_seed(41)
    shapes = [(64, 64, 3), (128, 512, 3), (24, 16, 1), (4, 64, 3), (70, 130, 3), (512, 256, 1)]
    ws = [torch.randn(co, ci, k, k, generator=g).cuda() for co, ci, k in shapes]
    extra = torch.randn(32, 8, 3, 3, generator=g).cuda()           # forward pack only: goes through the per-element kernel
    specs = [(w, 0, False) for w in ws] + [(extra, 0, True)] + [(w, 1, w.shape[0] <= 32 and w.shape[2] == 3) for w in ws]
    pb = ops.PackBatch(specs, prec)
    assert pb.npair == len(ws) and pb.nsingle == 1
    for pk in pb.packs:                                            # poison: padding must be written too
        pk.hi.fill_(0x7fff)
        if pk.lo is not None:
            pk.lo.fill_(0x7fff)
    got = pb.update()
    for (w, mode, sk), pk in zip(specs, got):
        ref = ops.pack_weights(w, mode, prec, small_k=sk)
        assert pk.hi.shape == ref.hi.shape and torch.equal(pk.hi, ref.hi), (tuple(w.shape), mode)
        if prec == 1:
            assert torch.equal(pk.lo, ref.lo), (tuple(w.shape), mode)


@pytest.mark.parametrize('relu_in', [False, True])
def test_l1_backward_from_the_saved_sign_pattern(relu_in):
    """the L1 forward can leave its 1-byte sign pattern; the backward from it equals the backward that re-reads a and b"""
    ops = _ops()
    g = torch.Generator().manual_seed(51)
    a = torch.randn(2, 16, 16, 64, generator=g).cuda()
    b = torch.randn(2, 16, 16, 64, generator=g).cuda()
    a[0, 0, 0, :8] = b[0, 0, 0, :8]                                 # exact ties -> zero gradient
    add = torch.randn(2, 16, 16, 64, generator=g).cuda()
    go = torch.tensor(0.37).cuda()
    term, sgn = ops.l1_sum(a, b, relu_in, 1.0 / a.numel(), want_sign=True)
    assert torch.equal(term, ops.l1_sum(a, b, relu_in, 1.0 / a.numel()))
    ref = ops.l1_bwd(a, b, go, 1.0 / a.numel(), relu_in, add=add)
    got = ops.l1_bwd(None, None, go, 1.0 / a.numel(), False, add=add, sign=sgn, shape=a.shape)
    assert torch.equal(got, ref)
    assert torch.equal(ops.l1_bwd(None, None, go, 2.0, False, sign=sgn, shape=a.shape), ops.l1_bwd(a, b, go, 2.0, relu_in))


@pytest.mark.parametrize('n,e,labels', [(98000, 512, [5, 97999, 5, 0, 40000, 5, 123, 0]), (7, 8, [3, 1])])
def test_label_embedding_gradient_kernel(n, e, labels):
    """lp_sn_embed_grad (the dense part of the spectrally normalised label embedding's gradient, nn.SNEmbeddingFn): rank-1 term + the B
    gradient rows added to an existing gradient, duplicate labels included, against the torch formulation it replaces"""
    from latent_pose_reenactment_amd import _lib
    g = torch.Generator().manual_seed(n + e)
    grad0 = torch.randn(n, e, generator=g).cuda()
    u, v = torch.randn(n, generator=g).cuda(), torch.randn(e, generator=g).cuda()
    coef = torch.tensor([0.37]).cuda()
    label = torch.tensor(labels).cuda()
    rows = torch.randn(len(labels), e, generator=g).cuda()
    want = grad0.double().clone()
    want.addmm_((u.double() * (-0.37))[:, None], v.double()[None, :])
    want.index_add_(0, label, rows.double())
    got = grad0.clone()
    _lib.check(_lib.lib().lp_sn_embed_grad(got.data_ptr(), u.data_ptr(), v.data_ptr(), coef.data_ptr(), label.data_ptr(), rows.data_ptr(), n, e,
                                           len(labels), torch.cuda.current_stream().cuda_stream), 'lp_sn_embed_grad')
    torch.cuda.synchronize()
    report(f'sn_embed_grad {n}x{e}', rel(got, want), 1e-6)
    for r in set(labels):
        assert rel(got[r], want[r]) < 1e-6, r


@pytest.mark.parametrize('prec', [2, 0])
@pytest.mark.parametrize('shape', [(2, 16, 16, 64), (1, 12, 20, 128), (3, 4, 4, 512)])
def test_plane_to_plane_pool_and_l1_against_planes(shape, prec):
    """lp_avgpool2_fwd16 (planes in, planes out) and lp_l1_fwd_b16 (second operand = 16-bit planes) -- the target-image half of the VGG
    criterions keeps no fp32 activation (criterions/common/perceptual_loss.py::_features16)"""
    ops = _ops()
    n, h, w, c = shape
    g = torch.Generator().manual_seed(5)
    dt = torch.float16 if prec == 2 else torch.bfloat16
    x = torch.randn(n, h, w, c, generator=g).cuda()
    a16 = ops.act_pack(x, pro=2, prec=prec)                       # planes of relu(x)
    dec = lambda t: t.view(dt).double()
    p16 = ops.avgpool2_fwd16(a16, prec)
    torch.cuda.synchronize()
    want = F.avg_pool2d(dec(a16.hi).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    report(f'avgpool2_fwd16{shape} prec={prec}', rel(dec(p16.hi), want), 2e-3 if prec == 2 else 1.6e-2)      # one rounding of the fp32 mean
    assert torch.equal(p16.hi.view(dt), want.to(dt)) or rel(dec(p16.hi), want.to(dt).double()) < 1e-3         # (RNE of the exact mean up to fp32 summation)
    other = torch.randn(n, h, w, c, generator=g).cuda()
    term, sgn = ops.l1_sum(other, ops.Tap16(a16, prec), True, 1.0 / other.numel(), want_sign=True)
    torch.cuda.synchronize()
    u, v = torch.relu(other.double()), dec(a16.hi)
    report(f'l1_fwd_b16{shape} prec={prec}', rel(term, (u - v).abs().mean()), 1e-5)
    ref_sgn = (torch.sign(u - v) * ((other > 0) | (u > v))).to(torch.int8).flatten()
    assert torch.equal(sgn, ref_sgn)
