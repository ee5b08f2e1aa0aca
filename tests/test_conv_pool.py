"""nn.ConvPoolFn (round 6): conv3x3(relu(h)) + bias, AvgPool2d(2), + res, ReLU as ONE 4x4 stride-2 conv launch (lp_pack_weights modes 4 / 5 on the phase
kernels) -- the critic's down blocks (discriminators/no_landmarks.py:52-81 of the reference via blocks.py:76-90).  Against fp64 autograd of the unfused
chain on the same 16-bit operand planes: output, data gradient (incl. the ReLU mask of relu(h)), weight / bias gradient, residual gradient."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize('prec', [1, 2])
@pytest.mark.parametrize('case', [(2, 32, 32, 64, 64, True), (3, 16, 24, 64, 128, True), (2, 64, 64, 128, 256, False), (8, 8, 8, 512, 512, True), (1, 256, 256, 64, 64, True)])
def test_conv_pool_forward_and_gradients(case, prec):
    from latent_pose_reenactment_amd import hipops as ops
    from latent_pose_reenactment_amd import nn as lpnn
    n, h, w, cin, cout, relu_out = case
    g = torch.Generator().manual_seed(sum(case[:5]) + prec)
    hh = torch.randn(n, h, w, cin, generator=g).cuda().requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda().requires_grad_(True)
    bias = torch.randn(cout, generator=g).cuda().requires_grad_(True)
    res = torch.randn(n, h // 2, w // 2, cout, generator=g).cuda().requires_grad_(True)
    r = torch.randn(n, h // 2, w // 2, cout, generator=g).cuda()
    x16 = ops.act_pack(hh.detach(), pro=2, prec=prec)          # planes of relu(h)
    holder = []
    y = lpnn.ConvPoolFn.apply(hh, wt, bias, res, prec, None, None, x16, relu_out, holder)
    (y * r).sum().backward()
    torch.cuda.synchronize()
    dt = torch.float16 if prec == 2 else torch.bfloat16
    A = x16.hi.view(dt).double()[..., :cin]
    if prec == 1:
        A = A + x16.lo.view(torch.bfloat16).double()[..., :cin]
    a64 = A.permute(0, 3, 1, 2).clone().requires_grad_(True)          # relu(h) as the kernel saw it
    w64, b64, r64 = wt.detach().double().requires_grad_(True), bias.detach().double().requires_grad_(True), res.detach().double().requires_grad_(True)
    y64 = F.avg_pool2d(F.conv2d(a64, w64, b64, 1, 1), 2) + r64.permute(0, 3, 1, 2)
    if relu_out:          # (tie-masked: the ReLU behind the pool on the kernel's own pattern -- a pooled value within rounding distance of 0 may flip)
        y64 = y64 * (y.detach().permute(0, 3, 1, 2) > 0)
    (y64 * r.double().permute(0, 3, 1, 2)).sum().backward()
    dh64 = a64.grad.permute(0, 2, 3, 1) * (hh.detach().double() > 0)          # the ReLU in front of the conv
    tol = {1: 3e-5, 2: 8e-4}[prec]
    errs = {'y': rel(y, y64.permute(0, 2, 3, 1)), 'dh': rel(hh.grad, dh64), 'dw': rel(wt.grad, w64.grad), 'db': rel(bias.grad, b64.grad), 'dres': rel(res.grad, r64.grad)}
    print(f'[conv-pool] prec={prec} {case}: ' + ' '.join(f'{k}={v:.2e}' for k, v in errs.items()))
    assert all(v < tol for v in errs.values()), errs
    o16 = holder[0]
    dec = o16.hi.view(dt).double()[..., :cout] + (o16.lo.view(torch.bfloat16).double()[..., :cout] if prec == 1 else 0)
    assert rel(dec, torch.relu(y.double()) if relu_out else y.double()) < {1: 1e-5, 2: 5e-4}[prec]
