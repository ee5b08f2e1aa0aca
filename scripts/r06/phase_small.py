import os, sys, itertools, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from latent_pose_reenactment_amd import hipops as ops
def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
bad = 0
for prec in (1, 2):
    for n, hl, cin, cout in itertools.product((1, 2, 3, 4), (4, 8, 16), (8, 16, 32, 64), (8, 16, 32, 128)):
        g = torch.Generator().manual_seed(n * 1000 + hl * 10 + cin + cout)
        x = torch.randn(n, hl, hl, cin, generator=g).cuda()
        dy = torch.randn(n, 2 * hl, 2 * hl, cout, generator=g).cuda()
        wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda()
        a = ops.act_pack(x, pro=2, prec=prec)
        d = ops.act_pack(dy, pro=0, prec=prec)
        ref = ops.conv16(a, ops.pack_weights(wt, 0, prec), ksize=3, upsample=True, prec=prec)
        got = ops.conv16(a, ops.pack_phase_weights(wt, prec), ksize=3, upsample=True, prec=prec, phase=True)
        ref_d = ops.sum2x2(ops.conv16(d, ops.pack_weights(wt, 1, prec), ksize=3, prec=prec))
        got_d = ops.conv16(d, ops.pack_phase_weights(wt, prec, dgrad=True), ksize=3, prec=prec, phase_dgrad=True)
        torch.cuda.synchronize()
        ef, ed = rel(got, ref), rel(got_d, ref_d)
        tol = 2e-5 if prec == 1 else 1e-3
        if not (ef < tol and ed < tol):
            bad += 1
            print(f'MISMATCH prec={prec} N={n} low-res {hl}x{hl} cin={cin} cout={cout}: forward {ef:.2e} dgrad {ed:.2e}', flush=True)
print('phase_small: mismatches', bad)
