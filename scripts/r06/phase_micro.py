"""round 6: the generator's x2-upsampled 3x3 convs (up1 .. up5, N = 8): fused-upsample kernel vs the phase forms -- forward, and the data gradient
(dense 3x3 data-gradient conv on the 2H x 2W grid + lp_sum2x2 vs ONE phase launch on the low-resolution grid).  PREC = 1 bf16x3 | 2 f16 | 0 bf16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from latent_pose_reenactment_amd import hipops as ops  # noqa: E402

prec = int(os.environ.get('PREC', '1'))
SHAPES = [(8, 16, 16, 512, 512), (8, 32, 32, 512, 512), (8, 64, 64, 512, 256), (8, 128, 128, 256, 128), (8, 256, 256, 128, 64)]      # N, Hout, Wout, Cin, Cout


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n, h, w, cin, cout in SHAPES:
    x = torch.randn(n, h // 2, w // 2, cin, device='cuda')
    dy = torch.randn(n, h, w, cout, device='cuda')
    wt = torch.randn(cout, cin, 3, 3, device='cuda') * 0.02
    a = ops.act_pack(x, pro=2, prec=prec)
    d = ops.act_pack(dy, prec=prec, grad=True)
    pf, pp = ops.pack_weights(wt, 0, prec), ops.pack_phase_weights(wt, prec)
    pt, pd = ops.pack_weights(wt, 1, prec), ops.pack_phase_weights(wt, prec, dgrad=True)
    fl = 2.0 * n * h * w * cin * cout * 9
    t_f = timeit(lambda: ops.conv16(a, pf, ksize=3, upsample=True, prec=prec))
    t_p = timeit(lambda: ops.conv16(a, pp, ksize=3, upsample=True, prec=prec, phase=True))
    t_d = timeit(lambda: ops.sum2x2(ops.conv16(d, pt, ksize=3, prec=prec)))
    t_q = timeit(lambda: ops.conv16(d, pd, ksize=3, prec=prec, phase_dgrad=True))
    print(f'prec={prec} {str((n, h, w, cin, cout)):28s} | forward: fused-upsample {t_f:6.1f} us -> phase {t_p:6.1f} us ({fl / t_p / 1e6:6.0f} TF/s dense count) '
          f'| data gradient: dense + sum2x2 {t_d:6.1f} us -> phase {t_q:6.1f} us ({fl / t_q / 1e6:6.0f} TF/s)', flush=True)
