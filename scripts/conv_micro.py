import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latent_pose_reenactment_amd import hipops as ops
SHAPES = [  # N, H, W, Cin, Cout, ks, ups
    (8, 4, 4, 512, 512, 3, 0), (8, 16, 16, 512, 512, 3, 0), (8, 32, 32, 512, 512, 3, 0), (8, 64, 64, 256, 256, 3, 0),
    (8, 128, 128, 128, 128, 3, 0), (8, 256, 256, 64, 64, 3, 0), (8, 256, 256, 128, 64, 3, 1), (8, 256, 256, 64, 128, 3, 0), (8, 256, 256, 3, 64, 3, 0), (8, 256, 256, 3, 64, 1, 0)]
prec = int(os.environ.get('PREC', '0'))
REPS = int(os.environ.get('REPS', '20'))
for (n, h, w, cin, cout, ks, ups) in SHAPES:
    hin, win = (h // 2, w // 2) if ups else (h, w)
    x = torch.randn(n, hin, win, cin, device='cuda')
    wgt = torch.randn(cout, cin, ks, ks, device='cuda') * 0.02
    sc = torch.randn(n, cin, device='cuda'); sh = torch.randn(n, cin, device='cuda')
    pack = ops.pack_weights(wgt, 0, prec, small_k=(ks == 3 and cin <= 32))
    pro = 0 if cin <= 4 else 1
    for _ in range(3):
        ops.conv(x, pack, ksize=ks, upsample=bool(ups), pro=pro, scale=sc, shift=sh, prec=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        ops.conv(x, pack, ksize=ks, upsample=bool(ups), pro=pro, scale=sc, shift=sh, prec=prec)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / REPS * 1e3
    fl = 2.0 * n * h * w * cin * cout * ks * ks
    alg = x.numel() * 4 + wgt.numel() * 2 * (2 if prec else 1) + n * h * w * cout * 4   # fp32 x read + packed W read + fp32 y write
    print(f'alg_bytes={alg} dbg={os.environ.get("LP_CONV_DBG","0"):>2s} cc={os.environ.get("LP_CONV_CC","-")} {str((n,h,w,cin,cout,ks,ups)):36s} {us:8.1f} us  {fl/us/1e6:7.1f} TF/s')
