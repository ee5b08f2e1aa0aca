"""GPU parity of the ResNeXt-50 identity encoder's HIP path (E1; embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:26,37-54):
  * every new kernel against its plain-torch fp64 restatement (tests/emu_ops.py) on the same inputs;
  * the whole network, forward AND backward, train- and eval-mode BatchNorm, against the stock nn.Module in fp64 on the same device.
Tolerances (rel-L2): bf16x3 3e-5 per op (fp32-class), f16 1e-3 per op (2^-12 operands); whole network: stated per test."""
import copy
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
pytestmark = pytest.mark.gpu

TOL = {0: 1e-2, 1: 3e-5, 2: 1e-3}


def _ops():
    from latent_pose_reenactment_amd import hipops
    return hipops


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def report(name, err, tol):
    print(f'[parity] {name}: rel-L2 {err:.3e} (tol {tol:.0e})')
    assert err < tol, f'{name}: rel-L2 {err:.3e} >= {tol}'


def decode(a, prec):
    """operand planes -> fp64 values (x 1/scale for scaled fp16 gradient operands)"""
    if prec == 2:
        v = a.hi.view(torch.float16).double()
    else:
        v = a.hi.view(torch.bfloat16).double()
        if prec == 1:
            v = v + a.lo.view(torch.bfloat16).double()
    if a.inv is not None:
        v = v * a.inv.double()
    return v[..., :a.c]


def e16(t):
    import emu_ops
    return emu_ops.Act16(t, None, t.shape[-1], None)


@pytest.mark.parametrize('prec', [1, 2])
def test_im2col_stem_rows(prec):
    import emu_ops
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    x = torch.rand(3, 3, 40, 56, generator=g).cuda()
    a = ops.im2col_planes(x, 7, 2, 3, prec)
    ref = emu_ops.im2col_planes(x.double(), 7, 2, 3, prec)
    assert a.hi.shape == (3, 20, 28, 152) and a.c == 147
    report(f'im2col 7x7/2 prec{prec}', rel(decode(a, prec), ref.hi), 3e-6 if prec == 1 else 3e-4)
    pad = a.hi[..., 147:]
    assert int(pad.abs().max()) == 0


@pytest.mark.parametrize('p,c', [(4096, 64), (1000, 256), (64, 2048), (70000, 128)])
def test_bn_train_stats(p, c):
    import emu_ops
    ops = _ops()
    g = torch.Generator().manual_seed(p + c)
    y = (torch.randn(p, c, generator=g) * torch.rand(c, generator=g) * 3 + torch.randn(c, generator=g) * 5).cuda()
    gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), torch.randn(c, generator=g).cuda()
    rm, rv = torch.randn(c, generator=g).cuda(), (torch.rand(c, generator=g) + 0.5).cuda()
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    out = ops.bn_train_stats(y, gamma, beta, rm, rv, 0.1, 1e-5)
    ref = emu_ops.bn_train_stats(y.double(), gamma.double(), beta.double(), rm_ref, rv_ref, 0.1, 1e-5)
    for nm, a, b in zip(('mean', 'rstd', 'scale', 'shift'), out, ref):
        report(f'bn_train_stats[{p}x{c}].{nm}', rel(a, b), 5e-6)
    report('running_mean', rel(rm, rm_ref), 1e-6)
    report('running_var', rel(rv, rv_ref), 5e-6)


@pytest.mark.parametrize('mode', ['relu', 'relu6', 'none', 'src', 'frozen'])
def test_norm_act_bwd_modes(mode):
    import emu_ops
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    n, h, w, c = 4, 12, 20, 64
    x = torch.randn(n, h, w, c, generator=g).cuda()
    dA = torch.randn(n, h, w, c, generator=g).cuda()
    src = torch.randn(n, h, w, c, generator=g).cuda()
    gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.3).cuda()
    mean, rstd, scale, shift = emu_ops.bn_train_stats(x.double(), gamma.double(), beta.double(), None, None, 0.1, 1e-5)
    if mode == 'relu6':
        scale, shift = scale * 4, shift * 4 + 2          # so that the upper clamp is active for a good share of the elements
    st32 = [t.float().contiguous() for t in (mean, rstd, scale, shift)]
    kw = dict(mask_mode={'relu': 0, 'relu6': 0, 'none': 1, 'src': 2, 'frozen': 0}[mode], mask_src=src if mode == 'src' else None,
              want_g=mode == 'src', act_hi=6.0 if mode == 'relu6' else 0.0, frozen=mode == 'frozen')
    dx, dg, db, gm = ops.norm_act_bwd(dA, x, gamma, *st32, amax=True, **kw)
    kw64 = dict(kw, mask_src=None if kw['mask_src'] is None else src.double())
    rdx, rdg, rdb, rg = emu_ops.norm_act_bwd(dA.double(), x.double(), gamma.double(), mean, rstd, scale, shift, **kw64)
    report(f'norm_act_bwd[{mode}].dx', rel(dx, rdx), 2e-5)
    report(f'norm_act_bwd[{mode}].dgamma', rel(dg, rdg), 2e-5)
    report(f'norm_act_bwd[{mode}].dbeta', rel(db, rdb), 2e-5)
    if gm is not None:
        report(f'norm_act_bwd[{mode}].g', rel(gm, rg), 1e-7)
    slots, ver = dx._lp_amax
    assert abs(float(slots.max()) - float(dx.abs().max())) <= 1e-6 * float(dx.abs().max())


@pytest.mark.parametrize('prec', [1, 2])
@pytest.mark.parametrize('mode', ['relu', 'relu6', 'none', 'src', 'frozen'])
def test_bn_bwd_straight_to_operand_planes(mode, prec):
    """lp_bn_bwd16: dy never exists in fp32 -- the planes (x 1/scale in fp16 mode) must equal the fp64 dy to operand precision, and the
    fp16 scale (taken from a per-channel BOUND of |dy|) must leave the largest element inside the fp16 range with >= 2^-3 of headroom used"""
    import emu_ops
    ops = _ops()
    g = torch.Generator().manual_seed(15)
    n, h, w, c = 4, 12, 20, 128
    x = torch.randn(n, h, w, c, generator=g).cuda()
    dA = (torch.randn(n, h, w, c, generator=g) * 1e-4).cuda()
    src = torch.randn(n, h, w, c, generator=g).cuda()
    gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.3).cuda()
    mean, rstd, scale, shift = emu_ops.bn_train_stats(x.double(), gamma.double(), beta.double(), None, None, 0.1, 1e-5)
    if mode == 'relu6':
        scale, shift = scale * 4, shift * 4 + 2
    st32 = [t.float().contiguous() for t in (mean, rstd, scale, shift)]
    kw = dict(mask_mode={'relu': 0, 'relu6': 0, 'none': 1, 'src': 2, 'frozen': 0}[mode], mask_src=src if mode == 'src' else None,
              want_g=mode == 'src', act_hi=6.0 if mode == 'relu6' else 0.0, frozen=mode == 'frozen')
    d16, dg, db, gm = ops.bn_bwd16(dA, x, gamma, *st32, prec=prec, **kw)
    kw64 = dict(kw, mask_src=None if kw['mask_src'] is None else src.double())
    rdx, rdg, rdb, rg = emu_ops.norm_act_bwd(dA.double(), x.double(), gamma.double(), mean, rstd, scale, shift, **kw64)
    report(f'bn_bwd16[{mode}] prec{prec} planes', rel(decode(d16, prec), rdx), 3e-6 if prec == 1 else 4e-4)
    report(f'bn_bwd16[{mode}] dgamma', rel(dg, rdg), 2e-5)
    report(f'bn_bwd16[{mode}] dbeta', rel(db, rdb), 2e-5)
    if gm is not None:
        report(f'bn_bwd16[{mode}] g', rel(gm, rg), 1e-7)
    if prec == 2:
        top = d16.hi.view(torch.float16).float().abs().max().item()
        assert 2.0 ** 9 <= top < 2.0 ** 13.01, top         # bound >= amax (no overflow) and within 2^4 of it (no needless underflow)


GCASES = [(2, 16, 16, 128, 4), (8, 8, 8, 1024, 32), (3, 12, 20, 256, 8), (1, 32, 32, 512, 16), (8, 4, 4, 1024, 32)]


@pytest.mark.parametrize('prec', [1, 2])
@pytest.mark.parametrize('case', GCASES)
def test_grouped_conv_fwd_dgrad_wgrad(case, prec):
    import emu_ops
    ops = _ops()
    n, h, w, c, cg = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, h, w, c, generator=g).cuda()
    wt = (torch.randn(c, cg, 3, 3, generator=g) / (cg * 9) ** 0.5).cuda()
    dy = (torch.randn(n, h, w, c, generator=g) * 1e-3).cuda()
    a = ops.act_pack(x, pro=0, prec=prec)
    y = ops.gconv16(a, ops.pack_grouped(wt, 0, prec), prec=prec)
    ry = emu_ops.gconv16(e16(x.double()), emu_ops.Pack(wt.double(), 0))
    report(f'gconv fwd {case} prec{prec}', rel(y, ry), TOL[prec])
    d16 = ops.act_pack(dy, prec=prec, grad=True)
    dx = ops.gconv16(d16, ops.pack_grouped(wt, 1, prec), prec=prec)
    rdx = emu_ops.gconv16(e16(dy.double()), emu_ops.Pack(wt.double(), 1))
    report(f'gconv dgrad {case} prec{prec}', rel(dx, rdx), TOL[prec])
    dw = ops.gconv_wgrad16(a, d16, cg, prec=prec)
    rdw = emu_ops.gconv_wgrad16(e16(x.double()), e16(dy.double()), cg)
    report(f'gconv wgrad {case} prec{prec}', rel(dw, rdw), TOL[prec])


@pytest.mark.parametrize('prec', [1, 2])
def test_grouped_conv_stride2_via_subsample_and_zero_stuffing(prec):
    """the stride-2 grouped conv (first block of layer2..4): full-resolution conv + pick; gradients through zero stuffing"""
    import torch.nn.functional as F
    ops = _ops()
    n, h, w, c, cg = 2, 16, 16, 256, 8
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, h, w, c, generator=g).cuda()
    wt = (torch.randn(c, cg, 3, 3, generator=g) / (cg * 9) ** 0.5).cuda()
    dy = torch.randn(n, h // 2, w // 2, c, generator=g).cuda()
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = wt.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, stride=2, padding=1, groups=c // cg)
    (yr * dy.double().permute(0, 3, 1, 2)).sum().backward()
    a = ops.act_pack(x, pro=0, prec=prec)
    y = ops.subsample2(ops.gconv16(a, ops.pack_grouped(wt, 0, prec), prec=prec))
    report(f'gconv stride 2 fwd prec{prec}', rel(y, yr.permute(0, 2, 3, 1)), TOL[prec])
    d16 = ops.zero_stuff2_16(ops.act_pack(dy, prec=prec, grad=True), h, w)
    dx = ops.gconv16(d16, ops.pack_grouped(wt, 1, prec), prec=prec)
    report(f'gconv stride 2 dgrad prec{prec}', rel(dx, xr.grad.permute(0, 2, 3, 1)), TOL[prec])
    dw = ops.gconv_wgrad16(a, d16, cg, prec=prec)
    report(f'gconv stride 2 wgrad prec{prec}', rel(dw, wr.grad), TOL[prec])


@pytest.mark.parametrize('prec', [1, 2])
def test_bn_relu_maxpool_fwd_bwd(prec):
    import emu_ops
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    n, h, w, c = 3, 22, 30, 64
    y = torch.randn(n, h, w, c, generator=g).cuda()
    sc, sh = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.3).cuda()
    out, o16, idx = ops.bn_relu_maxpool(y, sc, sh, prec)
    rout, _, ridx = emu_ops.bn_relu_maxpool(y.double(), sc.double(), sh.double(), prec)
    report('maxpool fwd', rel(out, rout), 1e-6)
    report('maxpool planes', rel(decode(o16, prec), rout), 3e-6 if prec == 1 else 3e-4)
    d = torch.randn(out.shape, generator=g).cuda()
    dA = ops.maxpool_bwd(d, idx, h, w)
    rdA = emu_ops.maxpool_bwd(d.double(), ridx, h, w)
    # ties at 0 (all-negative windows after the ReLU) may pick different positions; their gradient is zeroed by the ReLU backward anyway
    act = torch.relu(y.double() * sc.double() + sh.double()) > 0
    report('maxpool bwd (on active units)', rel(dA * act, rdA * act), 1e-6)


@pytest.mark.parametrize('prec', [1, 2])
@pytest.mark.parametrize('variant', ['identity', 'downsample', 'plain'])
def test_bn_add_act(variant, prec):
    import emu_ops
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    shp = (2, 9, 7, 256)
    y, res = torch.randn(shp, generator=g).cuda(), torch.randn(shp, generator=g).cuda()
    v = [(torch.rand(256, generator=g) + 0.5).cuda() if i % 2 == 0 else torch.randn(256, generator=g).cuda() for i in range(4)]
    args = dict(identity=(res, None, None), downsample=(res, v[2], v[3]), plain=(None, None, None))[variant]
    out, o16 = ops.bn_add_act(y, v[0], v[1], *args, relu=variant != 'plain', prec=prec)
    rout = emu_ops.bn_add_act(y.double(), v[0].double(), v[1].double(), *[None if t is None else t.double() for t in args], relu=variant != 'plain')
    report(f'bn_add_act[{variant}]', rel(out, rout), 1e-6)
    report(f'bn_add_act[{variant}] planes', rel(decode(o16, prec), rout), 3e-6 if prec == 1 else 3e-4)
    none, p16 = ops.bn_add_act(y, v[0], v[1], *args, relu=variant != 'plain', prec=prec, want_out=False)      # planes only (ABI 12): same planes, no fp32 out
    assert none is None and torch.equal(p16.hi, o16.hi) and (prec != 1 or torch.equal(p16.lo, o16.lo))


@pytest.mark.parametrize('prec', [0, 1])
def test_bn_add_act_with_the_residual_from_operand_planes(prec):
    """lp_bn_add_act_planes (ABI 12): the identity shortcut read from the operand planes of the block input equals lp_bn_add_act fed with the planes'
    decoded values BIT FOR BIT (fp32 out and planes), the planes-only form (no fp32 out) writes the same planes, and against the fp32 residual the
    result moves by the planes' own resolution (bf16x3: 2^-17 of the residual)"""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    shp = (2, 9, 7, 256)
    y, res = torch.randn(shp, generator=g).cuda(), torch.randn(shp, generator=g).cuda().relu()
    sc, sh = (torch.rand(256, generator=g) + 0.5).cuda(), torch.randn(256, generator=g).cuda()
    r16 = ops.act_pack(res, pro=0, prec=prec)
    rdec = decode(r16, prec).float().contiguous()
    out_a, p_a = ops.bn_add_act(y, sc, sh, r16, relu=True, prec=prec)
    out_b, p_b = ops.bn_add_act(y, sc, sh, rdec, relu=True, prec=prec)
    assert torch.equal(out_a, out_b) and torch.equal(p_a.hi, p_b.hi) and (prec == 0 or torch.equal(p_a.lo, p_b.lo))
    none, p_c = ops.bn_add_act(y, sc, sh, r16, relu=True, prec=prec, want_out=False)
    assert none is None and torch.equal(p_c.hi, p_a.hi) and (prec == 0 or torch.equal(p_c.lo, p_a.lo))
    out_f, _ = ops.bn_add_act(y, sc, sh, res, relu=True, prec=prec)
    report(f'bn_add_act_planes[{prec}] vs the fp32 residual', rel(out_a, out_f), 1e-5 if prec == 1 else 4e-3)


@pytest.mark.parametrize('mode', ['relu', 'src'])
def test_sixteen_bit_resident_conv_output_forms(mode):
    """fp16 mode, conv outputs 16-bit resident: (1) the conv / grouped-conv epilogue writes the fp16 plane of y (no fp32 y) and the SAME
    statistics partials as with fp32 y; (2) lp_bn_act16 / lp_bn_add_act16 / lp_bn_bwd16_h applied to that plane equal their fp32-input
    forms applied to the plane's decoded values EXACTLY where the arithmetic is the same (forward affine: bit-equal planes), and to
    summation order otherwise; (3) the ReLU pattern read from operand planes (mask mode 3) equals the fp32 mask."""
    import emu_ops
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    n, h, w, c = 4, 16, 16, 128
    a = ops.act_pack(torch.randn(n, h, w, c, generator=g).cuda(), pro=2, prec=2)
    wt = (torch.randn(c, 4, 3, 3, generator=g) * 0.1).cuda()
    pack = ops.pack_grouped(wt, 0, 2)
    y32, cs32 = ops.gconv16(a, pack, prec=2, stats=True)
    none, y16, cs16 = ops.gconv16(a, pack, prec=2, stats=True, want_y=False, out16=True)
    assert none is None and torch.equal(y16.hi.view(torch.float16), y32.half()) and torch.equal(cs16.part[:cs16.rows * n * c * 3], cs32.part[:cs32.rows * n * c * 3])
    fa = ops.flat16(a)
    w1 = (torch.randn(256, c, 1, 1, generator=g) * 0.1).cuda()
    p1 = ops.pack_weights(w1, 0, 2)
    z32, zs32 = ops.conv16(fa, p1, ksize=1, prec=2, stats=True)
    none, z16, zs16 = ops.conv16(fa, p1, ksize=1, prec=2, stats=True, want_y=False, out16=0)
    assert none is None and torch.equal(z16.hi.view(torch.float16), z32.half()) and zs16.rows == zs32.rows \
        and torch.equal(zs16.part[:zs16.rows * 256 * 3], zs32.part[:zs32.rows * 256 * 3])
    # consumers: y as the plane vs y as the plane's decoded fp32 values
    yd = ops.y16_to_f32(y16).contiguous()
    gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.3).cuda()
    mean, rstd, scale, shift = ops.norm_stats_finalize(cs16, 1, c, gamma, beta, 1e-5)
    pl16 = ops.bn_act16(y16, scale, shift)
    pl32 = ops.act_pack(yd, pro=4, scale=scale, shift=shift, prec=2)
    assert torch.equal(pl16.hi, pl32.hi)
    res = torch.randn(n, h, w, c, generator=g).cuda()
    o16, op16 = ops.bn_add_act(y16, scale, shift, res, relu=True, prec=2)
    o32, op32 = ops.bn_add_act(yd, scale, shift, res, relu=True, prec=2)
    assert torch.equal(o16, o32) and torch.equal(op16.hi, op32.hi)
    dA = (torch.randn(n, h, w, c, generator=g) * 1e-3).cuda()
    kw16 = dict(mask_mode=2, mask_src=op16, want_g=True) if mode == 'src' else {}
    kw32 = dict(mask_mode=2, mask_src=o32, want_g=True) if mode == 'src' else {}
    d16, dg16, db16, g16 = ops.bn_bwd16(dA, y16, gamma, mean, rstd, scale, shift, prec=2, **kw16)
    d32, dg32, db32, g32 = ops.bn_bwd16(dA, yd, gamma, mean, rstd, scale, shift, prec=2, **kw32)
    assert torch.equal(d16.hi, d32.hi) and torch.equal(d16.inv, d32.inv) and torch.equal(dg16, dg32) and torch.equal(db16, db32)
    if mode == 'src':
        assert torch.equal(g16, g32)
    rdx = emu_ops.norm_act_bwd(dA.double(), yd.double(), gamma.double(), mean.double(), rstd.double(), scale.double(), shift.double(),
                               mask_mode=kw32.get('mask_mode', 0), mask_src=None if mode != 'src' else o32.double())[0]
    report(f'bn_bwd16_h[{mode}] planes vs fp64', rel(decode(d16, 2), rdx), 4e-4)


def test_stride2_plumbing_and_pooling():
    import emu_ops
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(3, 10, 14, 64, generator=g).cuda()
    assert torch.equal(ops.subsample2(x), x[:, ::2, ::2].contiguous())
    a = ops.act_pack(x, pro=0, prec=1)
    s = ops.subsample2_16(a)
    assert torch.equal(s.hi, a.hi[:, ::2, ::2].contiguous()) and torch.equal(s.lo, a.lo[:, ::2, ::2].contiguous())
    z = ops.zero_stuff2_16(s, 10, 14)
    ref = torch.zeros_like(a.hi); ref[:, ::2, ::2] = s.hi
    assert torch.equal(z.hi, ref)
    d = torch.randn(3, 10, 14, 64, generator=g).cuda()
    sm = torch.randn(3, 5, 7, 64, generator=g).cuda()
    want = d.clone(); want[:, ::2, ::2] += sm
    assert torch.equal(ops.add_strided2(d, sm), want)
    report('spatial_mean', rel(ops.spatial_mean(x), x.double().mean(dim=(1, 2))), 1e-6)
    gm = torch.randn(3, 64, generator=g).cuda()
    report('spatial_mean_bwd', rel(ops.spatial_mean_bwd(gm, 10, 14), emu_ops.spatial_mean_bwd(gm.double(), 10, 14)), 1e-6)


@pytest.mark.parametrize('prec', [1, 2])
def test_bn_relu_pack_and_flat_1x1_contraction(prec):
    """conv -> BN -> ReLU -> conv as the embedder runs it: lp_act_pack pro 4 on a per-channel affine, then the 1x1 contraction on the
    pixels flattened to one image, its data gradient with a residual, and its weight gradient"""
    import emu_ops
    ops = _ops()
    g = torch.Generator().manual_seed(8)
    n, h, w, cin, cout = 8, 8, 8, 256, 128
    y = torch.randn(n, h, w, cin, generator=g).cuda()
    sc, sh = (torch.rand(cin, generator=g) + 0.5).cuda(), (torch.randn(cin, generator=g) * 0.3).cuda()
    wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).cuda()
    a = ops.act_pack(y, pro=4, scale=sc, shift=sh, prec=prec)
    ra = emu_ops.act_pack(y.double(), pro=4, scale=sc.double(), shift=sh.double())
    report('act_pack pro 4', rel(decode(a, prec), ra.hi), 3e-6 if prec == 1 else 3e-4)
    a5 = ops.act_pack(y, pro=5, scale=sc, shift=sh, prec=prec)
    report('act_pack pro 5', rel(decode(a5, prec), y.double() * sc.double() + sh.double()), 3e-6 if prec == 1 else 3e-4)
    fa = ops.flat16(a)
    assert fa.hi.shape == (1, 32, 16, cin)
    out = ops.conv16(fa, ops.pack_weights(wt, 0, prec), ksize=1, prec=prec).view(n, h, w, cout)
    rout = emu_ops.conv16(ra, emu_ops.Pack(wt.double(), 0), ksize=1)
    report('flat 1x1 fwd', rel(out, rout), TOL[prec])
    dy = (torch.randn(n, h, w, cout, generator=g) * 1e-2).cuda()
    res = torch.randn(n, h, w, cin, generator=g).cuda() * 1e-2
    d16 = ops.act_pack(dy, prec=prec, grad=True)
    dx = ops.conv16(ops.flat16(d16), ops.pack_weights(wt, 1, prec), ksize=1, res=res.view(1, 32, 16, cin), prec=prec).view(n, h, w, cin)
    rdx = emu_ops.conv16(e16(dy.double()), emu_ops.Pack(wt.double(), 1), ksize=1, res=res.double())
    report('flat 1x1 dgrad + res', rel(dx, rdx), TOL[prec])
    dw = ops.conv_wgrad16(fa, ops.flat16(d16), ksize=1, prec=prec)
    report('flat 1x1 wgrad', rel(dw, emu_ops.conv_wgrad16(ra, e16(dy.double()), ksize=1)), TOL[prec])


@pytest.mark.parametrize('prec', [0, 1, 2])
@pytest.mark.parametrize('p_cin_cout_splits', [(148, 24, 144, None), (4096, 64, 256, 8), (4096, 64, 256, 5), (2048, 320, 1280, 16),
                                              (1024, 2048, 1024, 2), (8192, 128, 128, 64), (60, 16, 8, 1)])
def test_pointwise_weight_gradient_kernel(prec, p_cin_cout_splits):
    """wgrad1x1_kernel (LDS-DMA staged 128 x 128 tiles, swizzled rows, XCD-ordered split-K) against the fp64 contraction of the SAME
    operand planes: only the fp32 accumulation order differs, so the gate is 1e-5 in every precision mode.  Ragged channel counts
    (MobileNetV2's 24 / 144 / 320 / 1280), pixel counts that are not a multiple of the 64-pixel stage, split counts with and without
    the XCD mapping (multiples of 8), more splits than stages."""
    ops = _ops()
    pix, cin, cout, splits = p_cin_cout_splits
    g = torch.Generator().manual_seed(pix + cin)
    fh, fw = ops.flat_hw(pix)
    x = torch.randn(1, fh, fw, cin, generator=g).cuda()
    dy = (torch.randn(1, fh, fw, cout, generator=g) * 1e-3).cuda()
    a = ops.act_pack(x, pro=2, prec=prec)
    d = ops.act_pack(dy, prec=prec, grad=True)
    dw = ops.conv_wgrad16(a, d, ksize=1, prec=prec, splits=splits)
    if prec == 1:       # bf16x3 multiplies hi*hi + hi*lo + lo*hi (the lo*lo term, 2^-16 relative, is dropped by design)
        pl = lambda t, c: t.view(torch.bfloat16).double().reshape(pix, -1)[:, :c]
        dh, dl, ah, al = pl(d.hi, cout), pl(d.lo, cout), pl(a.hi, cin), pl(a.lo, cin)
        ref = dh.t() @ ah + dh.t() @ al + dl.t() @ ah
    else:
        ref = decode(d, prec).reshape(pix, cout).t() @ decode(a, prec).reshape(pix, cin)
    report(f'wgrad1x1 P={pix} {cin}->{cout} splits={splits}', rel(dw.view(cout, cin), ref), 1e-5)


def _nets(num_classes, seed, layers=(3, 4, 6, 3)):
    from embedders.backbones import ResNeXt
    torch.manual_seed(seed)
    m = ResNeXt(list(layers), 32, 4, num_classes)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.uniform_(-0.3, 0.3)
            mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 1.5)
    return m.cuda(), copy.deepcopy(m).double().cuda()


def structured_frames(n, size, seed):
    """smooth, per-frame distinct content + mild noise in [0, 1]: white-noise frames make every deep feature almost constant over
    the batch, so that train-mode BatchNorm divides by a vanishing spread and ANY arithmetic (fp32 included) is amplified 1000-fold"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(n, 3, 8, 8, generator=g)
    x = F.interpolate(low, size=(size, size), mode='bilinear', align_corners=False)
    return (x + 0.1 * torch.rand(n, 3, size, size, generator=g)).clamp(0, 1)


def _grad_err(params, ref_params):
    num = sum((p.grad.double().cpu() - q.grad.double().cpu()).norm() ** 2 for p, q in zip(params, ref_params))
    den = sum(q.grad.double().cpu().norm() ** 2 for q in ref_params)
    return float((num / den) ** 0.5)


def _cos(params, ref_params):
    a = torch.cat([p.grad.double().cpu().reshape(-1) for p in params]); b = torch.cat([q.grad.double().cpu().reshape(-1) for q in ref_params])
    return float((a * b).sum() / (a.norm() * b.norm()))


@pytest.mark.parametrize('depth', ['shallow', 'resnext50'])
@pytest.mark.parametrize('prec_name,train', [('bf16x3', True), ('bf16x3', False), ('f16', True), ('f16', False)])
def test_resnext_forward_backward_vs_fp64(monkeypatch, prec_name, train, depth):
    """whole network through the HIP path vs the stock layers in fp64 (same device): logits, EVERY parameter gradient, BatchNorm buffers,
    with the stock fp32 layers (MIOpen / rocBLAS) against the same fp64 run printed as the calibration.
    A randomly initialised 50-layer ReLU network in train-mode BatchNorm with 8 frames is a chaotic map (shattered gradients: the stock
    fp32 layers themselves are 2e-2 off in the gradients), so the full-depth net only gets calibrated bounds; the SHALLOW variant
    (layers [2,1,1,1]: every kernel configuration -- group sizes 4/8/16/32, stride 1/2, identity and downsample blocks, stem, classifier --
    at a depth where arithmetic error is not amplified) carries the per-mode gates."""
    from oracle import backbones_ref as BR
    monkeypatch.setenv('LP_PREC_E', prec_name)
    size = 128
    m, ref = _nets(32, 7, (2, 1, 1, 1) if depth == 'shallow' else (3, 4, 6, 3))
    m32 = copy.deepcopy(m)
    for net in (m, ref, m32):
        net.train(train)
    x = structured_frames(8, size, 3).cuda()
    r = torch.randn(8, 32, device='cuda')
    y = m(x)
    assert m.__dict__.get('_hip_param_names') is not None, 'the HIP path did not run'
    (y * r).sum().backward()
    yr = BR.resnext_forward(ref, x.double())
    (yr * r.double()).sum().backward()
    y32 = BR.resnext_forward(m32, x)
    (y32 * r).sum().backward()
    berr = {k: rel(b.double(), q) for (k, b), (_, q) in zip(m.named_buffers(), ref.named_buffers()) if b.dtype.is_floating_point}
    e_out, e_b = rel(y, yr), max(berr.values())
    tot = _grad_err(list(m.parameters()), list(ref.parameters()))
    c_out, c_tot = rel(y32, yr), _grad_err(list(m32.parameters()), list(ref.parameters()))
    cos = _cos(list(m.parameters()), list(ref.parameters()))
    print(f'[parity] {depth} {prec_name} train={train} {size}px: logits {e_out:.2e}, all-gradients {tot:.2e} (cosine {cos:.4f}), buffers {e_b:.2e} '
          f'| stock fp32 layers vs fp64: logits {c_out:.2e}, all-gradients {c_tot:.2e}')
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    if depth == 'shallow':
        # gates = 2-3x what a CPU emulation of the SAME operand rounding predicts for this net and input (fp64 arithmetic, operands of every
        # contraction rounded to hi+lo bf16 / to fp16 with the power-of-two gradient scale: scripts/embedder_rounding_study.py):
        #   bf16x3: eval 6.7e-6 / 3.0e-4, train 3.6e-5 / 1.5e-2;   f16 (conv outputs 16-bit resident): eval 5.0e-4 / 1.7e-3, train 3.2e-3 / 1.4e-1
        #   (logits / all gradients; f16 with fp32-resident conv outputs, LP_E_Y16=0: 4.9e-4 / 1.7e-3 and 3.0e-3 / 1.3e-1)
        # i.e. the kernels reproduce the arithmetic they are specified to do; what is left is the conditioning of train-mode BatchNorm + ReLU
        # at random initialisation (the stock fp32 layers are 3e-3 off in the gradients on the same problem).
        tol = {('bf16x3', False): (2e-5, 1e-3, 1e-6), ('bf16x3', True): (1.5e-4, 5e-2, 3e-5),
               ('f16', False): (1.5e-3, 8e-3, 1e-6), ('f16', True): (1e-2, 0.3, 2e-3)}[(prec_name, train)]
        assert e_out < tol[0] and tot < tol[1] and e_b < tol[2], (e_out, tot, e_b, c_out, c_tot)
    elif prec_name == 'bf16x3':
        assert e_out < max(50 * c_out, 2e-5) and tot < max(10 * c_tot, 1e-3) and e_b < 1e-3, (e_out, tot, e_b, c_out, c_tot)
    else:
        assert e_out < 0.1 and cos > 0.5 and e_b < 2e-2, (e_out, cos, e_b)
    for (k, b), (_, q) in zip(m.named_buffers(), ref.named_buffers()):
        if not b.dtype.is_floating_point:
            assert int(b) == int(q), k


def test_embedder_plugin_uses_hip_identity_encoder(monkeypatch):
    """the plugin call site (get_identity_embedding: B x K frames -> embeds) reaches the HIP function, and the frame mean is differentiable"""
    import argparse
    from embedders.unsupervised_pose_separate_embResNeXt_segmentation import Wrapper as EW
    monkeypatch.setenv('LP_PREC_E', 'bf16x3')
    torch.manual_seed(0)
    E = EW.get_net(argparse.Namespace(embed_channels=16, pose_embedding_size=8, average_function='sum', device='cuda')).train()
    d = {'enc_rgbs': structured_frames(8, 128, 1).view(2, 4, 3, 128, 128).cuda()}
    E.get_identity_embedding(d)
    assert d['embeds'].shape == (2, 16) and d['embeds_elemwise'].shape == (2, 4, 16)
    d['embeds'].sum().backward()
    assert E.identity_encoder.__dict__.get('_hip_param_names') is not None
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in E.identity_encoder.parameters())


@pytest.mark.parametrize('n', [1, 3, 6, 13])
def test_eval_mode_takes_any_batch(monkeypatch, n):
    """ the reference's fine-tuning bootstrap (train.py:241-256: eval(), no_grad, b*k frames per batch) and few-shot runs hand the
    identity encoder 1 .. 7 frames or a count that is no multiple of 4; with running statistics the batch is zero-padded to the next count the
    kernels cover and the pad rows are dropped -- every frame's logits against the stock layers in fp64, with and without autograd.  The pose
    encoder likewise (autograd on: padded; no-grad: chunks of 64).  Train-mode BatchNorm keeps raising (the frames of a batch are coupled)."""
    from oracle import backbones_ref as BR
    monkeypatch.setenv('LP_PREC_E', 'bf16x3')
    m, ref = _nets(32, 7, (2, 1, 1, 1))
    m.eval(); ref.eval()
    x = structured_frames(n, 128, 3).cuda()
    with torch.no_grad():
        y0 = m(x)
    y = m(x)                      # autograd on (parameter gradients; the encoder does not differentiate w.r.t. its frames)
    assert y.shape == (n, 32) and torch.equal(y.detach(), y0)
    r = torch.randn(n, 32, device='cuda')
    (y * r).sum().backward()
    yr = BR.resnext_forward(ref, x.double())
    (yr * r.double()).sum().backward()
    e_out, e_g = rel(y, yr), _grad_err(list(m.parameters()), list(ref.parameters()))
    print(f'[parity] eval-mode identity encoder, {n} frames (padded to a covered batch): logits {e_out:.2e}, all-gradients {e_g:.2e}')
    assert e_out < 2e-5 and e_g < 1e-3, (e_out, e_g)
    m.train()
    with pytest.raises(RuntimeError, match='outside the HIP path'):
        m(x)
    from embedders import backbones
    torch.manual_seed(3)
    pe = backbones.mobilenet_v2(8).cuda().eval()
    pr = copy.deepcopy(pe).double()
    xp = structured_frames(n, 64, 5).cuda()
    yp = pe(xp)
    ypr = BR.mobilenet_forward(pr, xp.double())
    assert yp.shape == (n, 8) and rel(yp, ypr) < 1e-4, rel(yp, ypr)
