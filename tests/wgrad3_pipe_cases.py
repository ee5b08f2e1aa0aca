"""Cases of tests/test_wgrad3_pipe.py, run as a script in a fresh interpreter (LP_WGRAD3_PIPE = 2 puts the DMA-staged 3x3 weight-gradient
kernel of csrc/conv_wgrad.hip on every shape; the knob is read once per process).  Every case is checked against the fp64 contraction of
the SAME 16-bit operand planes (only the fp32 accumulation order differs: gate 2e-5 in both one-plane modes), bias gradient (column sums
of dy from the matrix core) included: ragged image sizes (partial 8 x 16 tiles: zero-page lanes), channel tails, several (co, ci) tiles,
one .. many pixel splits (with and without the XCD-aware block order), the fused x2 upsampling, the block-diagonal grouped form."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))
from latent_pose_reenactment_amd import hipops as ops  # noqa: E402


def dec(t, prec, c):
    return t.view(torch.float16 if prec == 2 else torch.bfloat16).double()[..., :c].permute(0, 3, 1, 2)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


CASES = [  # N, H, W, Cin, Cout, ups, splits (None: default)
    (1, 8, 16, 64, 64, 0, None),            # one tile, one split
    (2, 32, 32, 64, 64, 0, None),           # 16 tiles
    (2, 64, 64, 64, 64, 0, 16),             # 128 tiles in 16 splits (XCD-aware order), 8 tiles per workgroup
    (2, 64, 64, 64, 64, 0, 5),              # 5 splits (plain order), uneven ranges
    (1, 40, 24, 96, 160, 0, None),          # ragged in both directions, channel tails (CiP 128, CoP 192)
    (3, 16, 48, 128, 256, 0, 4),            # 2 x 4 (ci, co) tiles
    (2, 32, 32, 64, 128, 1, None),          # fused x2 upsampling
    (1, 48, 80, 160, 64, 1, 3),             # upsampled, ragged, channel tail
    (8, 4, 4, 64, 128, 0, None),            # tiny maps (forced mode only): half-empty tiles
    (2, 12, 10, 72, 40, 0, None),           # W < 16, odd channel counts
    (1, 128, 128, 64, 64, 0, None),         # 128 tiles, default splits
]

GROUPED = [  # N, H, W, C, group size
    (2, 16, 16, 128, 4),
    (1, 32, 16, 256, 8),
    (3, 8, 24, 64, 16),
    (2, 16, 32, 128, 32),                   # groups of 32: two row fragments per wave
    (1, 40, 24, 192, 4),                    # ragged tiles, three 64-channel blocks
]


def run_case(case, prec):
    n, h, w, cin, cout, ups, splits = case
    g = torch.Generator().manual_seed(sum(case[:6]) + prec)
    hin, win = (h // 2, w // 2) if ups else (h, w)
    x = torch.randn(n, hin, win, cin, generator=g).cuda()
    dy = torch.randn(n, h, w, cout, generator=g).cuda()
    a = ops.act_pack(x, pro=2, prec=prec)
    d = ops.act_pack(dy, pro=0, prec=prec)
    dw, db = ops.conv_wgrad16(a, d, ksize=3, upsample=bool(ups), prec=prec, splits=splits, bias_grad=True)
    torch.cuda.synchronize()
    A, D = dec(a.hi, prec, cin), dec(d.hi, prec, cout)
    if prec == 1:          # bf16x3: operands = hi + lo (the kernel drops the lo x lo term: 2^-16 relative)
        A, D = A + dec(a.lo, prec, cin), D + dec(d.lo, prec, cout)
    if ups:
        A = A.repeat_interleave(2, 2).repeat_interleave(2, 3)
    wgt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device=A.device, requires_grad=True)
    F.conv2d(A, wgt, None, 1, 1).backward(D)
    return {'dw': rel(dw, wgt.grad), 'db': rel(db, D.sum(dim=(0, 2, 3)))}


def run_grouped(case, prec):
    n, h, w, c, cg = case
    g = torch.Generator().manual_seed(sum(case) + prec)
    x = torch.randn(n, h, w, c, generator=g).cuda()
    dy = torch.randn(n, h, w, c, generator=g).cuda()
    a = ops.act_pack(x, pro=2, prec=prec)
    d = ops.act_pack(dy, pro=0, prec=prec)
    dw = ops.gconv_wgrad16(a, d, cg, prec=prec)
    torch.cuda.synchronize()
    A, D = dec(a.hi, prec, c), dec(d.hi, prec, c)
    if prec == 1:          # bf16x3: operands = hi + lo (the kernel drops the lo x lo term: 2^-16 relative)
        A, D = A + dec(a.lo, prec, c), D + dec(d.lo, prec, c)
    wgt = torch.zeros(c, cg, 3, 3, dtype=torch.float64, device=A.device, requires_grad=True)
    F.conv2d(A, wgt, None, 1, 1, 1, c // cg).backward(D)
    return {'dw': rel(dw, wgt.grad)}


def main():
    bad = []
    for prec in (2, 0):
        for case in CASES:
            if os.environ.get('LP_WGRAD3_PIPE') != '2' and (case[2] < 16 or case[1] < 8):
                continue
            errs = run_case(case, prec)
            print(f'[wgrad3_pipe mode={os.environ.get("LP_WGRAD3_PIPE")}] prec={prec} {case}: ' + ' '.join(f'{k}={v:.2e}' for k, v in errs.items()), flush=True)
            bad += [(prec, case, k, v) for k, v in errs.items() if not v < 2e-5]
        for case in GROUPED:
            errs = run_grouped(case, prec)
            print(f'[wgrad3_pipe grouped] prec={prec} {case}: ' + ' '.join(f'{k}={v:.2e}' for k, v in errs.items()), flush=True)
            bad += [(prec, case, k, v) for k, v in errs.items() if not v < 2e-5]
    if os.environ.get('LP_WGRAD3_X3', '1') != '0':          # (round 6) the dense bf16x3 layers on the DMA-staged kernel (one workgroup per CU, 512 registers)
        for case in CASES:
            if os.environ.get('LP_WGRAD3_PIPE') != '2' and (case[2] < 16 or case[1] < 8):
                continue
            errs = run_case(case, 1)
            print(f'[wgrad3_pipe dense bf16x3 mode={os.environ.get("LP_WGRAD3_PIPE")}] {case}: ' + ' '.join(f'{k}={v:.2e}' for k, v in errs.items()), flush=True)
            bad += [(1, case, k, v) for k, v in errs.items() if not v < 3e-5]
    for case in GROUPED:                     # bf16x3 grouped (block-diagonal) forms
        errs = run_grouped(case, 1)
        print(f'[wgrad3_pipe grouped] prec=1 {case}: ' + ' '.join(f'{k}={v:.2e}' for k, v in errs.items()), flush=True)
        bad += [(1, case, k, v) for k, v in errs.items() if not v < 3e-5]
    if bad:
        print('FAILED', bad)
        sys.exit(1)
    print('WGRAD3_PIPE_OK')


if __name__ == '__main__':
    main()
