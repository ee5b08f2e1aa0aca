"""GPU parity of the norm statistics a conv epilogue leaves behind (lp_conv16_fwd_stats / lp_gconv16_fwd_stats + lp_norm_stats_finalize)
against the two-launch statistics kernels run on the written fp32 output (lp_instnorm_stats / lp_bn_train_stats, themselves pinned by the
golden fixtures) and against fp64: instance norm per (n, c) for the generator's AdaIN (generators/common/blocks.py:18-26) and BatchNorm
over all pixels for the embedder; plus the geometries that must fall back (maps under 64 pixels, split-K launches) and the fp32-free
output mode (y = NULL, operand planes only)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))
pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


CASES = [  # N, H, W, Cin, Cout, ks, ups, res
    (8, 32, 32, 128, 128, 3, 0, 0), (8, 64, 64, 64, 64, 3, 0, 1), (3, 32, 32, 64, 128, 3, 1, 0), (2, 128, 128, 64, 64, 3, 0, 0),
    (5, 16, 32, 64, 192, 3, 0, 0), (8, 16, 16, 128, 64, 1, 0, 0), (2, 64, 64, 128, 64, 3, 1, 1), (1, 8, 8, 256, 128, 3, 0, 0)]


@pytest.mark.parametrize('prec', [1, 2])
@pytest.mark.parametrize('case', CASES)
def test_conv_epilogue_instance_norm_statistics(case, prec):
    from latent_pose_reenactment_amd import hipops as ops
    n, h, w, cin, cout, ks, ups, has_res = case
    g = torch.Generator().manual_seed(sum(case))
    hin, win = (h // 2, w // 2) if ups else (h, w)
    x = (torch.randn(n, hin, win, cin, generator=g) + 0.5).cuda()
    wt = (torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5).cuda()
    res = (torch.randn(n, h, w, cout, generator=g) * 2 + 3).cuda() if has_res else None        # a large mean: the shifted sums matter
    gamma, beta = (torch.rand(n, cout, generator=g) + 0.5).cuda(), torch.randn(n, cout, generator=g).cuda()
    a = ops.act_pack(x, pro=0, prec=prec)
    y, cs = ops.conv16(a, ops.pack_weights(wt, 0, prec), ksize=ks, upsample=bool(ups), res=res, prec=prec, stats=True)
    want = ops.instnorm_stats(y, gamma, beta, 1e-4)
    y64 = y.double().reshape(n, h * w, cout)
    mean64, var64 = y64.mean(1), y64.var(1, unbiased=False)
    if cs is None:
        assert n * h * w <= 8192 and ks == 3, f'the fused statistics should cover {case}'      # only launches small enough to run split-K
        return
    assert cs.rows == h * w // 64
    got = ops.norm_stats_finalize(cs, n, cout, gamma, beta, 1e-4)
    for nm, a_, b_ in zip(('mean', 'rstd', 'scale', 'shift'), got, want):
        e = rel(a_, b_)
        print(f'[parity] conv-epilogue statistics {case} prec{prec} {nm}: {e:.2e} vs the statistics kernel')
        assert e < 5e-6, (nm, e)
    assert rel(got[0], mean64) < 3e-6 and rel(got[1], (var64 + 1e-4).rsqrt()) < 3e-6


@pytest.mark.parametrize('prec', [1, 2])
def test_flat_1x1_and_grouped_batchnorm_statistics(prec):
    """the embedder's usage: BatchNorm over ALL pixels (one 'image' of P pixels) with the running-statistics update"""
    from latent_pose_reenactment_amd import hipops as ops
    g = torch.Generator().manual_seed(9)
    n, h, w, cin, cout = 8, 16, 16, 256, 128
    x = (torch.randn(n, h, w, cin, generator=g) + 0.3).cuda()
    wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).cuda()
    gamma, beta = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
    a = ops.act_pack(x, pro=0, prec=prec)
    y, cs = ops.conv16(ops.flat16(a), ops.pack_weights(wt, 0, prec), ksize=1, prec=prec, stats=True)
    assert cs is not None and cs.rows == n * h * w // 64 and cs.images == 1
    rm, rv = torch.zeros(cout).cuda(), torch.ones(cout).cuda()
    rm2, rv2 = rm.clone(), rv.clone()
    got = ops.norm_stats_finalize(cs, 1, cout, gamma, beta, 1e-5, running_mean=rm, running_var=rv, momentum=0.1)
    want = ops.bn_train_stats(y, gamma, beta, rm2, rv2, 0.1, 1e-5)
    for nm, a_, b_ in zip(('mean', 'rstd', 'scale', 'shift', 'running_mean', 'running_var'), got + (rm, rv), want + (rm2, rv2)):
        e = rel(a_, b_)
        print(f'[parity] flat 1x1 BatchNorm statistics prec{prec} {nm}: {e:.2e}')
        assert e < 5e-6, (nm, e)
    # grouped 3x3
    c, cg = 256, 8
    xg = torch.randn(4, 16, 16, c, generator=g).cuda()
    wg = (torch.randn(c, cg, 3, 3, generator=g) / (cg * 9) ** 0.5).cuda()
    ag = ops.act_pack(xg, pro=0, prec=prec)
    yg, csg = ops.gconv16(ag, ops.pack_grouped(wg, 0, prec), prec=prec, stats=True)
    assert csg is not None and csg.rows == 4 and csg.images == 4
    gam, bet = torch.ones(c).cuda(), torch.zeros(c).cuda()
    got = ops.norm_stats_finalize(csg, 1, c, gam, bet, 1e-5)
    want = ops.bn_train_stats(yg, gam, bet, None, None, 0.1, 1e-5)
    for nm, a_, b_ in zip(('mean', 'rstd'), got, want):
        assert rel(a_, b_) < 5e-6, (nm, rel(a_, b_))


@pytest.mark.parametrize('prec', [1, 2])
def test_planes_only_output(prec):
    """y = NULL: the conv writes only the consumer's operand planes (16-bit activation residency) -- same planes, same statistics"""
    from latent_pose_reenactment_amd import hipops as ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(4, 32, 32, 64, generator=g).cuda()
    wt = (torch.randn(128, 64, 3, 3, generator=g) / 24).cuda()
    bias = torch.randn(128, generator=g).cuda()
    a = ops.act_pack(x, pro=0, prec=prec)
    pk = ops.pack_weights(wt, 0, prec)
    y, o16, cs = ops.conv16(a, pk, ksize=3, bias=bias, prec=prec, out16=1, stats=True)
    y2, o16b, cs2 = ops.conv16(a, pk, ksize=3, bias=bias, prec=prec, out16=1, stats=True, want_y=False)
    assert y2 is None and torch.equal(o16.hi, o16b.hi) and (prec != 1 or torch.equal(o16.lo, o16b.lo))
    assert cs is not None and cs2 is not None and torch.equal(cs.part[:cs.rows * 4 * 128 * 3], cs2.part[:cs.rows * 4 * 128 * 3])
