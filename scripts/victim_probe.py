"""Which kernels give different results when they run BESIDE the LDS-DMA conv kernels on another stream?  (round 5: the spectral-norm power
iteration did -- scripts/sn_determinism.py SN_NOISE=conv16 -- until spectral_norm.hip was built without packed fp32 arithmetic:
profiles/r05_pk_fp32_opsel_hazard.txt.)  Victims: instance-norm statistics (LDS reduction), act_pack (no LDS), the power
iteration, a conv; noise: lp_conv16_fwd on 8 x 64 x 64 x 256 -> 256 (conv_pipe_kernel, or conv_dma_kernel with LP_CONV_PIPE=0).
Part 2 names the stage of the power iteration that parts first and takes one disturbed element apart.
usage: python scripts/victim_probe.py [repeats=60]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from latent_pose_reenactment_amd import hipops as ops  # noqa: E402
from latent_pose_reenactment_amd.nn import SNBatch, SNWeight  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.manual_seed(0)
x = torch.randn(8, 64, 64, 256, device='cuda')
w = torch.randn(256, 256, 3, 3, device='cuda') * 0.02
prec = {'f16': 2, 'bf16': 0, 'bf16x3': 1}[os.environ.get('LP_PREC', 'f16')]
pk = ops.pack_weights(w, 0, prec)
a = ops.act_pack(x, pro=0, prec=prec)
xv = torch.randn(8, 32, 32, 512, device='cuda')
layers = [SNWeight((512, 512, 3, 3), False, 1e-4).cuda() for _ in range(4)]
snb = SNBatch(layers)
init = [(l.weight_u.clone(), l.weight_v.clone()) for l in layers]
side = torch.cuda.Stream()


def victims():
    st = ops.instnorm_stats(xv, None, None, 1e-4)
    pl = ops.act_pack(xv, pro=2, prec=2)
    for l, (u, v) in zip(layers, init):
        l.weight_u.copy_(u); l.weight_v.copy_(v)
    with torch.no_grad():
        snb.update(True)
    y = ops.conv16(ops.act_pack(xv, pro=0, prec=2), ops.pack_weights(torch.ones(512, 512, 3, 3, device='cuda') * 0.01, 0, 2), ksize=3, prec=2)
    return {'instnorm': torch.stack(st).clone(), 'act_pack': pl.hi.clone(), 'power_iter': torch.cat([l.weight_u for l in layers]).clone(), 'conv_32x32': y.clone()}


ref = victims()
torch.cuda.synchronize()
bad = {}
for i in range(reps):
    noisy = i % 2 == 1
    if noisy:
        with torch.cuda.stream(side):
            for _ in range(6):
                ops.conv16(a, pk, ksize=3, prec=prec)
    cur = victims()
    torch.cuda.synchronize()
    for k in ref:
        same = torch.equal(ref[k].view(torch.int32) if ref[k].dtype == torch.float32 else ref[k], cur[k].view(torch.int32) if cur[k].dtype == torch.float32 else cur[k])
        if not same:
            bad.setdefault(k, [0, 0])[1 if noisy else 0] += 1
print(f'[victims] LP_CONV_PIPE={os.environ.get("LP_CONV_PIPE", "1")} LP_PREC={os.environ.get("LP_PREC", "f16")}: {reps // 2} quiet + {reps // 2} noisy runs; mismatching runs [quiet, noisy] per victim: {bad if bad else "none"}')

# ---- which stage of the power iteration parts first?  One buffer set (SETS = 1): after a quiet and after a noisy update from the same
# (W, u, v) the set's intermediates are compared: part rows (sn_wtu), v_out (sn_vsum + sn_v), s (sn_wv), u_out (sn_u)
SNBatch.SETS = 1
snb1 = SNBatch(layers)


def one(noisy):
    for l, (u, v) in zip(layers, init):
        l.weight_u.copy_(u); l.weight_v.copy_(v)
    torch.cuda.synchronize()
    if noisy:
        with torch.cuda.stream(side):
            for _ in range(6):
                ops.conv16(a, pk, ksize=3, prec=prec)
    with torch.no_grad():
        snb1.update(True)
    torch.cuda.synchronize()
    sig, uo, vo, scratch = snb1.sets[0][2]
    return scratch.clone(), vo.clone(), uo.clone(), sig.clone()


q = one(False)
rows, cols = 512, 4608
nrb = (rows + 31) // 32
need = ((nrb * cols + rows + (cols + 63) // 64) + 3) // 4 * 4
for trial in range(6):
    n = one(True)
    msgs = []
    for li in range(len(layers)):
        sc_q, sc_n = q[0][li * need:(li + 1) * need], n[0][li * need:(li + 1) * need]
        part_q, part_n = sc_q[:nrb * cols].view(nrb, cols), sc_n[:nrb * cols].view(nrb, cols)
        bad_rows = [int(r) for r in torch.nonzero((part_q != part_n).any(dim=1))[:, 0]]
        s_bad = int((sc_q[nrb * cols:nrb * cols + rows] != sc_n[nrb * cols:nrb * cols + rows]).sum())
        nrm_bad = int((sc_q[nrb * cols + rows:nrb * cols + rows + 72] != sc_n[nrb * cols + rows:nrb * cols + rows + 72]).sum())
        v_bad = int((q[1][li * cols:(li + 1) * cols] != n[1][li * cols:(li + 1) * cols]).sum())
        u_bad = int((q[2][li * rows:(li + 1) * rows] != n[2][li * rows:(li + 1) * rows]).sum())
        if bad_rows or s_bad or v_bad or u_bad or nrm_bad:
            cols_bad = []
            if bad_rows:
                r0 = bad_rows[0]
                cb = torch.nonzero(part_q[r0] != part_n[r0])[:, 0]
                cols_bad = [int(cb.min()), int(cb.max()), int(cb.numel()), float((part_q[r0] - part_n[r0]).abs().max()), float(part_q[r0].abs().max())]
            msgs.append(f'layer {li}: W^T u partial rows differing {bad_rows[:8]} (first row: cols {cols_bad}); norm partials {nrm_bad}; v_out {v_bad}; W v rows {s_bad}; u_out {u_bad}')
    print(f'[stages] noisy trial {trial}: ' + ('identical to the quiet run' if not msgs else ' | '.join(msgs)), flush=True)

# ---- anatomy of one disturbed element: quiet - noisy against the 32 single-row contributions W[r][c] * u[r] of its row block
W0 = layers[0].weight_orig.detach().reshape(512, -1).double()
u0 = init[0][0].double()
for trial in range(3):
    n = one(True)
    part_q, part_n = q[0][:nrb * cols].view(nrb, cols), n[0][:nrb * cols].view(nrb, cols)
    idx = torch.nonzero(part_q != part_n)
    print(f'[anatomy] trial {trial}: {idx.shape[0]} differing elements in layer 0', flush=True)
    for rb, c in idx[:6].tolist():
        t = (W0[rb * 32:(rb + 1) * 32, c] * u0[rb * 32:(rb + 1) * 32])
        dq = float(part_q[rb, c]) - float(part_n[rb, c])
        exact = float(t.sum())
        best_row = int((t - dq).abs().argmin()); g4 = t.view(8, 4).sum(1); best_g = int((g4 - dq).abs().argmin())
        print(f'   rb {rb} col {c} (thread {(c // 4) % 256}, comp {c % 4}): quiet {float(part_q[rb, c]):+.6e} noisy {float(part_n[rb, c]):+.6e} exact {exact:+.6e} | quiet-noisy {dq:+.3e}; '
              f'nearest single row term r={best_row}: {float(t[best_row]):+.3e}; nearest 4-row group {best_g}: {float(g4[best_g]):+.3e}', flush=True)
