"""Is lp_sn_power_iter bitwise repeatable?  The critic's spectrally normalised layers (full size) take three power iterations from the SAME
(W, u, v) state many times over -- alone, and beside a bandwidth-heavy kernel on another stream -- and the resulting (u, v, sigma) bit patterns
are compared with the first run.  (Two data-parallel ranks sharing one GPU showed different u / v for identical weights: scripts/dp_replica_diag.py.)
usage: python scripts/sn_determinism.py [repeats=30]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))
import torch  # noqa: E402

import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
args = bench.make_args(256, 8, 'cuda:0', 1, 0, 'f16', finetune=False)
torch.manual_seed(123)
from discriminators.no_landmarks import Wrapper as DW  # noqa: E402
D = DW.get_net(args).train()
layers = D._conv_sn_layers()
names = {id(m): k for k, m in D.named_modules()}
init = [(l.weight_u.clone(), l.weight_v.clone()) for l in layers]
side = torch.cuda.Stream()
junk = torch.empty(64 * 1024 * 1024, device='cuda')
NOISE = os.environ.get('SN_NOISE', 'mul')          # mul: an element-wise kernel on the side stream; conv: the generator's forward (LDS-DMA conv kernels)
from latent_pose_reenactment_amd import hipops as ops  # noqa: E402
if NOISE in ('conv16', 'conv16small', 'actpack', 'instnorm', 'linear', 'gsn'):
    xs_ = torch.randn(8, 64, 64, 256, device='cuda')
    w_ = torch.randn(256, 256, 3, 3, device='cuda') * 0.02
    pk_ = ops.pack_weights(w_, 0, 2)
    a_ = ops.act_pack(xs_, pro=0, prec=2)
    xsm_ = torch.randn(8, 8, 8, 512, device='cuda')
    wsm_ = torch.randn(512, 512, 3, 3, device='cuda') * 0.02
    pksm_ = ops.pack_weights(wsm_, 0, 2)
    asm_ = ops.act_pack(xsm_, pro=0, prec=2)
    xl_, wl_, bl_ = torch.randn(8, 768, device='cuda'), torch.randn(13056, 768, device='cuda') * 0.02, torch.zeros(13056, device='cuda')
    from latent_pose_reenactment_amd.nn import SNBatch as _SNB, SNWeight as _SNW  # noqa: E402
    glayers_ = [_SNW((512, 512, 3, 3), False, 1e-4).cuda() for _ in range(6)]
    gsn_ = _SNB(glayers_)
    torch.cuda.synchronize()


def noise_kernels():
    if NOISE == 'conv16':
        for _ in range(4):
            ops.conv16(a_, pk_, ksize=3, prec=2)
    elif NOISE == 'conv16small':
        for _ in range(8):
            ops.conv16(asm_, pksm_, ksize=3, prec=2)
    elif NOISE == 'actpack':
        for _ in range(8):
            ops.act_pack(xs_, pro=0, prec=2)
    elif NOISE == 'instnorm':
        for _ in range(8):
            ops.instnorm_stats(xs_, None, None, 1e-4)
    elif NOISE == 'linear':
        for _ in range(8):
            ops.linear_fwd(xl_, wl_, bl_, None)
    elif NOISE == 'gsn':
        with torch.no_grad():
            for _ in range(4):
                gsn_.update(True)


if NOISE == 'conv':
    from generators.vector_pose_unsupervised_segmentation_noBottleneck import Wrapper as GW  # noqa: E402
    Gn = GW.get_net(args).eval()
    dn = {'embeds': torch.randn(8, 512, device='cuda'), 'pose_embedding': torch.randn(8, 256, device='cuda')}
    with torch.no_grad():
        Gn(dict(dn))
    torch.cuda.synchronize()


def run(noise):
    for l, (u, v) in zip(layers, init):
        l.weight_u.copy_(u); l.weight_v.copy_(v)
    torch.cuda.synchronize()
    if noise:
        with torch.cuda.stream(side):
            if NOISE == 'conv':
                with torch.no_grad():
                    Gn(dict(dn))
            elif NOISE != 'mul':
                noise_kernels()
            else:
                for _ in range(4):
                    junk.mul_(1.0001)
    sig = []
    for _ in range(3):
        st = D._sn_batch.update(True)
        sig.append(torch.stack([s[2][:2].clone() for s in st]))
    torch.cuda.synchronize()
    return [(l.weight_u.clone(), l.weight_v.clone()) for l in layers], torch.stack(sig)


ref, sref = run(False)
bad = {}
for i in range(reps):
    cur, scur = run(i % 2 == 1)
    for l, (u0, v0), (u1, v1) in zip(layers, ref, cur):
        if not (torch.equal(u0.view(torch.int32), u1.view(torch.int32)) and torch.equal(v0.view(torch.int32), v1.view(torch.int32))):
            k = names[id(l)]
            bad[k] = bad.get(k, 0) + 1
    if not torch.equal(sref.view(torch.int32), scur.view(torch.int32)):
        bad['sigma'] = bad.get('sigma', 0) + 1
print(f'[sn-determinism] {reps} repeats of 3 power iterations over {len(layers)} layers: ' + ('bitwise repeatable' if not bad else f'MISMATCHES {bad}'))

# ---- part 2: hidden-state dependence.  W CHANGES between power iterations (as it does between training steps); every iteration runs twice from the
# same (W, u, v): on the long-lived SNBatch (rotating buffer sets holding the previous iterations' scratch) and on a brand-new SNBatch (zeroed buffers).
from latent_pose_reenactment_amd.nn import SNBatch  # noqa: E402
g = torch.Generator(device='cuda').manual_seed(5)
bad2 = {}
old = D._sn_batch
for it in range(reps):
    with torch.no_grad():
        for l in layers:
            l.weight_orig.add_(torch.randn(l.weight_orig.shape, device='cuda', generator=g) * (2e-4))
    keep = [(l.weight_u.clone(), l.weight_v.clone()) for l in layers]
    old.update(True)
    torch.cuda.synchronize()
    a = [(l.weight_u.clone(), l.weight_v.clone()) for l in layers]
    for l, (u, v) in zip(layers, keep):
        l.weight_u.copy_(u); l.weight_v.copy_(v)
    fresh = SNBatch(layers)
    fresh.update(True)
    torch.cuda.synchronize()
    for l, (u0, v0) in zip(layers, a):
        if not (torch.equal(u0.view(torch.int32), l.weight_u.view(torch.int32)) and torch.equal(v0.view(torch.int32), l.weight_v.view(torch.int32))):
            k = names[id(l)]
            d_ = float((u0 - l.weight_u).abs().max())
            bad2[k] = (bad2.get(k, (0, 0.0))[0] + 1, max(bad2.get(k, (0, 0.0))[1], d_))
print(f'[sn-determinism] long-lived SNBatch vs a fresh one over {reps} iterations with W changing in between: ' + ('bitwise identical' if not bad2 else f'MISMATCHES (count, max |du|) {bad2}'))
