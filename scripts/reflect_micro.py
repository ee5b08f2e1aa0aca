"""Per-shape time of the three reflection-padding border kernels (csrc/reflect_border.hip) on the generator's / critic's layer shapes.
usage: python scripts/reflect_micro.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from latent_pose_reenactment_amd import hipops as ops

SHAPES = [(8, 4, 512, 512, False), (8, 8, 512, 512, True), (8, 16, 512, 512, True), (8, 32, 512, 512, True), (8, 64, 512, 256, True), (8, 64, 256, 256, False),
          (8, 128, 256, 128, True), (8, 128, 128, 128, False), (8, 256, 128, 64, True), (8, 256, 64, 64, False), (8, 128, 64, 128, False), (8, 64, 128, 256, False)]
prec = 2


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f'{"N x H x W, Cin -> Cout, up":34s} {"fwd us":>8s} {"dgrad us":>9s} {"wgrad us":>9s}')
for n, h, cin, cout, up in SHAPES:
    hs = h // 2 if up else h
    x = torch.randn(n, hs, hs, cin, device='cuda')
    a = ops.act_pack(x, pro=2, prec=prec)
    w = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
    y = torch.zeros(n, h, h, cout, device='cuda')
    dy = torch.randn(n, h, h, cout, device='cuda')
    dx = torch.zeros(n, h, h, cin, device='cuda')
    alpha = torch.ones(1, device='cuda')
    tf = timed(lambda: ops.reflect_border_fwd(a, w, alpha, y, prec=prec, upsample=up))
    td = timed(lambda: ops.reflect_border_dgrad(dy, w, alpha, dx))
    tw = timed(lambda: ops.reflect_border_wgrad(a, dy, prec=prec, upsample=up))
    print(f'{n} x {h} x {h}, {cin:3d} -> {cout:3d}, up={int(up)}          {tf:8.1f} {td:9.1f} {tw:9.1f}', flush=True)
