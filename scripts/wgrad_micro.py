import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latent_pose_reenactment_amd import hipops as ops
SHAPES = [  # N, H, W, Cin, Cout, ks, ups
    (8, 4, 4, 512, 512, 3, 0), (8, 16, 16, 512, 512, 3, 0), (8, 32, 32, 512, 512, 3, 0), (8, 64, 64, 256, 256, 3, 0),
    (8, 128, 128, 128, 128, 3, 0), (8, 256, 256, 64, 64, 3, 0), (8, 256, 256, 128, 64, 3, 1), (8, 256, 256, 64, 4, 3, 0), (8, 256, 256, 3, 64, 3, 0), (8, 256, 256, 3, 64, 1, 0)]
prec = int(os.environ.get('PREC', '0'))
for (n, h, w, cin, cout, ks, ups) in SHAPES:
    hin, win = (h // 2, w // 2) if ups else (h, w)
    x = torch.randn(n, hin, win, cin, device='cuda'); dy = torch.randn(n, h, w, cout, device='cuda')
    sc = torch.randn(n, cin, device='cuda'); sh = torch.randn(n, cin, device='cuda')
    pro = 0 if cin <= 4 else 1
    f = lambda: ops.conv_wgrad(x, dy, ksize=ks, upsample=bool(ups), pro=pro, scale=sc, shift=sh, prec=prec)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * n * h * w * cin * cout * ks * ks
    print(f'wgrad prec={prec} {str((n,h,w,cin,cout,ks,ups)):36s} {us:8.1f} us  {fl/us/1e6:7.1f} TF/s')
