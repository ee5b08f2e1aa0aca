"""Phase-decomposed x2-upsampled 3x3 conv (round 6; csrc/conv_dma.hip KS = 2 form, hipops.pack_phase_weights / conv16(..., phase=True)): output pixel
(2y + a, 2x + b) of conv3x3(nearest_up2(x)) only meets 2 x 2 low-resolution pixels, the coinciding taps are summed beforehand -- 4/9 of the matrix work
of the fused-upsample kernel (SURVEY 7 "hard parts": exact algebra).  Checked against the fp64 conv of the SAME 16-bit operand planes over the
upsampled input with the ORIGINAL 3x3 weights (so the pre-summed weights' own 16-bit rounding is inside the tolerance: 2^-9 bf16, 2^-12 fp16, 2^-17
bf16x3), and against the fused-upsample kernel itself; every epilogue option: bias, low-resolution residual, 1/sigma, planes of y, norm statistics."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

CASES = [  # N, Hout, Wout, Cin, Cout, bias, res, out16, stats
    (8, 32, 32, 512, 512, 0, 1, None, 1),          # up2 of the generator: 16 x 16 -> 32 x 32
    (8, 64, 64, 512, 256, 0, 1, None, 1),
    (8, 128, 128, 256, 128, 1, 1, 0, 1),
    (8, 256, 256, 128, 64, 0, 1, None, 1),         # 64-channel tiles (256 x 64)
    (2, 48, 32, 160, 256, 1, 0, 1, 0),             # ragged low-resolution grid (24 x 16), five chunks
    (1, 16, 16, 64, 128, 0, 1, None, 1),           # 8 x 8 input: several images' worth of tile rows empty
    (3, 8, 8, 96, 136, 1, 0, None, 0),             # 4 x 4 input, channel tails
    (3, 16, 16, 32, 16, 0, 1, None, 1),            # 8 x 8 input, three images: two images per tile -> the statistics must fall back (or be right)
    (4, 8, 8, 32, 32, 0, 0, 0, 1),
]


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize('prec', [1, 2, 0])
@pytest.mark.parametrize('case', CASES)
def test_phase_conv_matches_the_fused_upsample_conv(case, prec):
    from latent_pose_reenactment_amd import hipops as ops
    n, h, w, cin, cout, has_bias, has_res, out16, stats = case
    g = torch.Generator().manual_seed(sum(case[:5]) + prec)
    x = torch.randn(n, h // 2, w // 2, cin, generator=g).cuda()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda()
    bias = torch.randn(cout, generator=g).cuda() if has_bias else None
    res = torch.randn(n, h // 2, w // 2, cout, generator=g).cuda() if has_res else None
    alpha = torch.tensor([0.83], device='cuda')
    a = ops.act_pack(x, pro=2, prec=prec)
    kw = dict(ksize=3, upsample=True, bias=bias, res=res, res_shift=1 if has_res else 0, alpha=alpha, prec=prec, out16=out16, stats=bool(stats))
    ref = ops.conv16(a, ops.pack_weights(wt, 0, prec), **kw)
    got = ops.conv16(a, ops.pack_phase_weights(wt, prec), phase=True, **kw)
    ref, got = (ref if isinstance(ref, tuple) else (ref,)), (got if isinstance(got, tuple) else (got,))
    torch.cuda.synchronize()
    dt = torch.float16 if prec == 2 else torch.bfloat16
    xs = a.hi.view(dt).double()[..., :cin]
    if prec == 1:
        xs = xs + a.lo.view(torch.bfloat16).double()[..., :cin]
    up = F.interpolate(xs.permute(0, 3, 1, 2), scale_factor=2, mode='nearest')
    y64 = F.conv2d(up, wt.double(), None, 1, 1).permute(0, 2, 3, 1) * 0.83
    if has_bias:
        y64 = y64 + bias.double()
    if has_res:
        y64 = y64 + res.double().repeat_interleave(2, 1).repeat_interleave(2, 2)
    tol = {1: 2e-5, 2: 6e-4, 0: 6e-3}[prec]          # the weights' operand rounding (original taps on one side, pre-summed taps on the other)
    e_ref, e_got = rel(ref[0], y64), rel(got[0], y64)
    print(f'[phase-conv] prec={prec} {case}: fused-upsample kernel {e_ref:.2e}, phase form {e_got:.2e} vs the fp64 conv of the planes')
    assert e_got < tol and e_got < 4 * e_ref + 1e-6, (e_got, e_ref)
    k = 1
    if out16 is not None:
        p_ref, p_got = ref[k], got[k]; k += 1
        dec = lambda t: t.hi.view(dt).double()[..., :cout] + (t.lo.view(torch.bfloat16).double()[..., :cout] if prec == 1 else 0)
        want = torch.relu(got[0].double()) if out16 else got[0].double()
        assert rel(dec(p_got), want) < {1: 1e-5, 2: 5e-4, 0: 4e-3}[prec]          # the planes are those of THIS launch's y
    if stats:
        s_ref, s_got = ref[k], got[k]
        assert (s_ref is None) == (s_got is None) or s_got is None, 'the phase form covers no more geometries than it says'
        if s_got is None and stats:
            pass
        if s_got is not None:          # instance-norm statistics from the partials of the phase launch == those of y itself
            gamma, beta = torch.ones(n, cout, device='cuda'), torch.zeros(n, cout, device='cuda')
            m, r, _, _ = ops.norm_stats_finalize(s_got, n, cout, gamma, beta, 1e-4)
            y = got[0].double()
            assert rel(m, y.mean((1, 2))) < 1e-5 and rel(r, 1 / torch.sqrt(y.var((1, 2), unbiased=False) + 1e-4)) < 1e-5


DGRAD_CASES = [  # N, Hlo, Wlo, Cin (of the forward conv = channels of dx), Cout (channels of dy)
    (8, 16, 16, 512, 512), (8, 32, 32, 512, 256), (8, 64, 64, 256, 128), (8, 128, 128, 128, 64),
    (2, 24, 16, 160, 256),          # ragged tiles
    (1, 8, 8, 64, 128), (3, 4, 4, 96, 136),
]


@pytest.mark.parametrize('prec', [1, 2, 0])
@pytest.mark.parametrize('case', DGRAD_CASES)
def test_phase_form_of_the_data_gradient(case, prec):
    """dx_lo = sum2x2( conv3x3^T(dy) ) -- the data gradient of conv3x3(nearest_up2(x)) w.r.t. the low-resolution x -- from ONE launch over the
    low-resolution grid (K loop over 4 phases x 2 x 2 taps, stride-2 gather of the dy planes): against fp64 autograd of the same planes and
    against the two-step form (dense 3x3 data-gradient conv on the 2H x 2W grid + lp_sum2x2)"""
    from latent_pose_reenactment_amd import hipops as ops
    n, h, w, cin, cout = case
    g = torch.Generator().manual_seed(sum(case) + prec)
    dy = torch.randn(n, 2 * h, 2 * w, cout, generator=g).cuda()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cout * 9) ** 0.5).cuda()
    alpha = torch.tensor([1.21], device='cuda')
    d16 = ops.act_pack(dy, pro=0, prec=prec)
    got = ops.conv16(d16, ops.pack_phase_weights(wt, prec, dgrad=True), ksize=3, alpha=alpha, prec=prec, phase_dgrad=True)
    two = ops.sum2x2(ops.conv16(d16, ops.pack_weights(wt, 1, prec), ksize=3, alpha=alpha, prec=prec))
    torch.cuda.synchronize()
    assert got.shape == (n, h, w, cin)
    dt = torch.float16 if prec == 2 else torch.bfloat16
    D = d16.hi.view(dt).double()[..., :cout]
    if prec == 1:
        D = D + d16.lo.view(torch.bfloat16).double()[..., :cout]
    if d16.inv is not None:          # (fp16 gradient operands carry a power-of-two scale; conv16 undoes it through alpha2)
        D = D * d16.inv.double()
    x = torch.zeros(n, cin, h, w, dtype=torch.float64, device='cuda', requires_grad=True)
    F.conv2d(F.interpolate(x, scale_factor=2, mode='nearest'), wt.double(), None, 1, 1).backward(D.permute(0, 3, 1, 2))
    ref = x.grad.permute(0, 2, 3, 1) * 1.21
    e_two, e_got = rel(two, ref), rel(got, ref)
    print(f'[phase-dgrad] prec={prec} {case}: dense data gradient + 2x2 sum {e_two:.2e}, phase form {e_got:.2e} vs fp64 autograd of the planes')
    assert e_got < {1: 2e-5, 2: 6e-4, 0: 6e-3}[prec] and e_got < 4 * e_two + 1e-6, (e_got, e_two)
