"""Reflection padding as a border correction of the zero-padded conv (csrc/reflect_border.hip; reference: nn.ReflectionPad2d(1) in front of the
ResBlocks' 3x3 convs, generators/common/blocks.py:76-88 with --gen_padding / --dis_padding reflection).  The three entry points are compared
with fp64 autograd of  conv(reflect_pad(x)) - conv(zero_pad(x))  on the operand values the kernels decode (the 16-bit planes): forward terms,
their transpose onto the ring one pixel inside the border (with and without the ReLU mask of a fused prologue), and the border's share of
the weight gradient -- plain, behind a x2 nearest upsample, ragged channel counts, every operand mode.  Module-level parity against the
reference's own outputs: tests/test_generator_module.py and tests/test_discriminator_criterions.py with the *_reflection fixtures."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

CASES = [   # (N, H, W, Cin, Cout, upsample)
    (2, 4, 4, 16, 16, False), (3, 8, 8, 24, 40, False), (2, 8, 8, 16, 8, True), (8, 16, 16, 64, 64, False), (1, 4, 8, 8, 72, False),
    (9, 8, 4, 70, 12, False), (2, 32, 32, 32, 32, True), (2, 64, 64, 8, 8, False),
]
PRECS = [('bf16', 0), ('bf16x3', 1), ('f16', 2)]


def _decode(a, prec):
    from latent_pose_reenactment_amd import hipops as ops
    hi = a.hi[..., :a.c]
    if prec == 2:
        return hi.view(torch.float16).double()
    v = (hi.to(torch.int32) << 16).view(torch.float32).double()
    if prec == 1:
        v = v + (a.lo[..., :a.c].to(torch.int32) << 16).view(torch.float32).double()
    return v


def _diff_conv(xt, w, up):
    """conv(reflect_pad(x~)) - conv(zero_pad(x~)), x~ = up2?(xt); NCHW fp64"""
    if up:
        xt = F.interpolate(xt, scale_factor=2, mode='nearest')
    return F.conv2d(F.pad(xt, (1, 1, 1, 1), mode='reflect'), w) - F.conv2d(xt, w, padding=1)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize('pname,prec', PRECS)
@pytest.mark.parametrize('case', CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_border_terms_forward_data_gradient_weight_gradient(case, pname, prec):
    from latent_pose_reenactment_amd import hipops as ops
    n, h, w_, cin, cout, up = case
    torch.manual_seed(sum(case) + prec)
    hs, ws = (h // 2, w_ // 2) if up else (h, w_)
    x = torch.randn(n, hs, ws, cin, device='cuda')
    wgt = (torch.randn(cout, cin, 3, 3, device='cuda') * 0.2).contiguous()
    alpha = torch.tensor([1.7, 0.6], device='cuda')[1:]
    a = ops.act_pack(x, pro=2, prec=prec)                                   # relu(x) planes: also the mask of the fused-prologue case
    xt = _decode(a, prec).permute(0, 3, 1, 2).requires_grad_(True)          # NCHW fp64, exactly what the kernels read
    wd = wgt.double().requires_grad_(True)
    ref = _diff_conv(xt, wd * 0.6, up)                                      # [n, cout, h, w]
    # forward: += into an existing tensor
    base = torch.randn(n, h, w_, cout, device='cuda')
    y = base.clone()
    ops.reflect_border_fwd(a, wgt, alpha, y, prec=prec, upsample=up)
    got = (y - base).permute(0, 3, 1, 2)
    assert rel(got, ref.detach()) < 2e-6, ('fwd', rel(got, ref.detach()))
    interior = got[:, :, 1:-1, 1:-1]
    assert float(interior.abs().max()) == 0.0, 'the correction touched a pixel inside the border'
    # backward of the difference
    dy = torch.randn(n, h, w_, cout, device='cuda')
    gx, gw = torch.autograd.grad(ref, (xt, wd), dy.permute(0, 3, 1, 2).double())
    gw_got = ops.reflect_border_wgrad(a, dy, prec=prec, upsample=up)
    assert rel(gw_got, gw / 0.6) < 2e-6, ('wgrad', rel(gw_got, gw / 0.6))          # raw gradient w.r.t. W * alpha (the SN rule scales it)
    assert float(gw_got[:, :, 1, 1].abs().max()) == 0.0
    if not up:       # the data-gradient correction works at the conv's resolution (an up block's x2 sum happens downstream)
        for masked in (False, True):
            dbase = torch.randn(n, h, w_, cin, device='cuda')
            dx = dbase.clone()
            ops.reflect_border_dgrad(dy, wgt, alpha, dx, a if masked else None)
            want = gx.permute(0, 2, 3, 1)
            if masked:
                want = want * (_decode(a, prec) > 0)
            assert rel(dx - dbase, want) < 2e-6, ('dgrad', masked, rel(dx - dbase, want))
    else:
        dxf = torch.zeros(n, h, w_, cin, device='cuda')
        ops.reflect_border_dgrad(dy, wgt, alpha, dxf)
        want = F.avg_pool2d(dxf.permute(0, 3, 1, 2).double(), 2) * 4            # sum over each 2 x 2 group = gradient w.r.t. the half-resolution tensor
        assert rel(want, gx) < 2e-6, ('dgrad-up', rel(want, gx))


def test_bad_geometry_is_refused():
    from latent_pose_reenactment_amd import hipops as ops
    x = torch.randn(1, 2, 2, 8, device='cuda')
    a = ops.act_pack(x, pro=0, prec=1)
    with pytest.raises(RuntimeError, match='bad geometry'):
        ops.reflect_border_fwd(a, torch.zeros(8, 8, 3, 3, device='cuda'), None, torch.zeros(1, 2, 2, 8, device='cuda'), prec=1)
