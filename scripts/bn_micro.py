"""Micro-benchmark of the BatchNorm-side kernels of the embedder on its own tensor shapes (ResNeXt-50 32x4d over 8 frames of 256 px:
P pixels x C channels, NHWC fp32): lp_bn_bwd16 (partial + finalize + apply), lp_bn_train_stats, lp_norm_act_bwd, lp_act_pack (BN + ReLU
prologue), lp_bn_add_act.  Prints us and GB/s of ALGORITHMIC bytes.  Knobs: LP_STAT_SPLIT_FIXED=1 (fixed 1024-pixel stage-1 split),
LP_BNB_ITEMS=1 (one-item-per-thread apply), PREC."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latent_pose_reenactment_amd import hipops as ops
SHAPES = [(131072, 64), (32768, 128), (32768, 256), (8192, 256), (8192, 512), (2048, 512), (2048, 1024), (512, 1024), (512, 2048)]
prec = int(os.environ.get('PREC', '2'))
REPS = int(os.environ.get('REPS', '30'))
pb = 6 if prec == 1 else 4          # bytes written per element of operand planes + 0


def timeit(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


for p, c in SHAPES:
    h, w = ops.flat_hw(p)
    x = torch.randn(1, h, w, c, device='cuda')
    dA = torch.randn(1, h, w, c, device='cuda')
    gamma = torch.rand(c, device='cuda') + 0.5
    beta = torch.randn(c, device='cuda') * 0.1
    mean, rstd, scale, shift = ops.bn_train_stats(x, gamma, beta, None, None, 0.1, 1e-5)
    n = p * c
    row = [f'P={p:6d} C={c:4d}']
    us = timeit(lambda: ops.bn_bwd16(dA, x, gamma, mean, rstd, scale, shift, prec=prec))
    row.append(f'bn_bwd16 {us:6.1f} us {n * (16 + pb - 2) / us / 1e3:5.0f} GB/s')
    us = timeit(lambda: ops.bn_train_stats(x, gamma, beta, None, None, 0.1, 1e-5))
    row.append(f'bn_stats {us:6.1f} us {n * 4 / us / 1e3:5.0f} GB/s')
    us = timeit(lambda: ops.norm_act_bwd(dA, x, gamma, mean, rstd, scale, shift))
    row.append(f'norm_act_bwd {us:6.1f} us {n * 20 / us / 1e3:5.0f} GB/s')
    us = timeit(lambda: ops.act_pack(x, pro=4, scale=scale, shift=shift, prec=prec))
    row.append(f'act_pack {us:6.1f} us {n * (4 + pb - 2) / us / 1e3:5.0f} GB/s')
    us = timeit(lambda: ops.bn_add_act(x, scale, shift, dA, relu=True, prec=prec))
    row.append(f'bn_add_act {us:6.1f} us {n * (12 + pb - 2) / us / 1e3:5.0f} GB/s')
    print(' | '.join(row), flush=True)
