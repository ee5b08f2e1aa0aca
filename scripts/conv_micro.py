"""Per-layer micro-benchmark of lp_conv16_fwd (operand planes already packed: the conv kernel alone), lp_act_pack and
lp_conv16_wgrad at the generator / critic / VGG layer classes, N = 8.  PREC = 0 bf16 | 1 bf16x3 | 2 f16; knobs LP_CONV_PP,
LP_CONV_W8, LP_CONV_KSPLIT, LP_WGRAD_COB.  Prints us and ALGORITHMIC TF/s (2*MACs of the dense conv / time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latent_pose_reenactment_amd import hipops as ops
SHAPES = [  # N, H, W, Cin, Cout, ks, ups
    (8, 4, 4, 512, 512, 3, 0), (8, 16, 16, 512, 512, 3, 0), (8, 32, 32, 512, 512, 3, 0), (8, 64, 64, 256, 256, 3, 0),
    (8, 128, 128, 128, 128, 3, 0), (8, 256, 256, 64, 64, 3, 0), (8, 256, 256, 128, 64, 3, 1), (8, 256, 256, 64, 128, 3, 0),
    (8, 64, 64, 512, 256, 3, 1), (8, 64, 64, 256, 128, 1, 0), (16, 128, 128, 128, 128, 3, 0), (16, 64, 64, 256, 256, 3, 0)]
if os.environ.get('SHAPES') == 'ups':        # the generator's x2-upsampled 3x3 convs (up1 .. up5); PHASE=1: their phase-decomposed form (round 6)
    SHAPES = [(8, 16, 16, 512, 512, 3, 1), (8, 32, 32, 512, 512, 3, 1), (8, 64, 64, 512, 256, 3, 1), (8, 128, 128, 256, 128, 3, 1), (8, 256, 256, 128, 64, 3, 1)]
PHASE = os.environ.get('PHASE', '0') != '0'
if os.environ.get('SHAPES') == 'small':      # the latency-bound layer classes (4x4 .. 16x16 maps, 1x1 skips)
    SHAPES = [(8, 4, 4, 512, 512, 3, 0), (8, 8, 8, 512, 512, 3, 0), (8, 16, 16, 512, 512, 3, 0), (8, 8, 8, 512, 512, 1, 0),
              (8, 16, 16, 512, 512, 1, 0), (8, 8, 8, 512, 512, 3, 1), (8, 32, 32, 512, 512, 3, 0)]
if os.environ.get('SHAPES') == '1x1':        # the embedder's pointwise layers on flattened pixels (ResNeXt-50 32x4d, 64 frames of 256 px)
    SHAPES = [(1, p // 16, 16, ci, co, 1, 0) for p, ci, co in
              [(262144, 64, 128), (262144, 128, 256), (262144, 256, 128), (65536, 256, 512), (65536, 512, 256), (16384, 512, 1024),
               (16384, 1024, 512), (4096, 1024, 2048), (4096, 2048, 1024)]]
if os.environ.get('SHAPES') == 'wgrad':      # the 3x3 weight-gradient classes of the meta-training step (generator, critic)
    SHAPES = [(8, 256, 256, 64, 64, 3, 0), (8, 128, 128, 64, 128, 3, 0), (8, 128, 128, 128, 128, 3, 0), (8, 64, 64, 128, 256, 3, 0),
              (8, 64, 64, 256, 256, 3, 0), (8, 32, 32, 256, 512, 3, 0), (8, 32, 32, 512, 512, 3, 0), (8, 16, 16, 512, 512, 3, 0),
              (8, 256, 256, 128, 64, 3, 1), (8, 128, 128, 256, 128, 3, 1), (8, 64, 64, 512, 256, 3, 1), (8, 32, 32, 512, 512, 3, 1)]
BIAS = os.environ.get('BIAS', '0') != '0'     # the weight-gradient launch also produces the bias gradient (as the step's layers do)
prec = int(os.environ.get('PREC', '0'))
REPS = int(os.environ.get('REPS', '20'))
WHAT = os.environ.get('WHAT', 'conv,pack,wgrad').split(',')


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


for (n, h, w, cin, cout, ks, ups) in SHAPES:
    hin, win = (h // 2, w // 2) if ups else (h, w)
    x = torch.randn(n, hin, win, cin, device='cuda')
    dy = torch.randn(n, h, w, cout, device='cuda')
    wgt = torch.randn(cout, cin, ks, ks, device='cuda') * 0.02
    sc = torch.randn(n, cin, device='cuda'); sh = torch.randn(n, cin, device='cuda')
    pack = ops.pack_phase_weights(wgt, prec) if (PHASE and ups and ks == 3) else ops.pack_weights(wgt, 0, prec)
    a = ops.act_pack(x, pro=1, scale=sc, shift=sh, prec=prec)
    d = ops.act_pack(dy, prec=prec, grad=True)
    fl = 2.0 * n * h * w * cin * cout * ks * ks
    out = [f'prec={prec} {str((n, h, w, cin, cout, ks, ups)):40s}']
    if 'conv' in WHAT:
        us = timeit(lambda: ops.conv16(a, pack, ksize=ks, upsample=bool(ups), prec=prec, phase=bool(PHASE and ups and ks == 3)))
        out.append(f'conv {us:7.1f} us {fl / us / 1e6:7.1f} TF/s')
    if 'pack' in WHAT:
        us = timeit(lambda: ops.act_pack(x, pro=1, scale=sc, shift=sh, prec=prec))
        gb = x.numel() * (4 + (4 if prec == 1 else 2)) / 1e3
        out.append(f'pack {us:6.1f} us {gb / us:6.0f} GB/s')
    if 'wgrad' in WHAT:
        us = timeit(lambda: ops.conv_wgrad16(a, d, ksize=ks, upsample=bool(ups), prec=prec, bias_grad=BIAS))
        out.append(f'wgrad {us:7.1f} us {fl / us / 1e6:7.1f} TF/s')
    print(' | '.join(out), flush=True)
