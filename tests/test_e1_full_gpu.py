"""Identity encoder (torchvision resnext50_32x4d(num_classes=512), embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:
26,37-54) at the FULL configs[2] geometry -- 8 samples x 8 frames of 256 x 256, train-mode BatchNorm, the initialisation bench.py uses -- forward
AND backward through the HIP encoder in the precision assignment the benchmark runs, against the stock layers of oracle/backbones_ref.py in
fp64 on the same device, with the stock fp32 layers (the reference's own arithmetic class) against the same fp64 run as the calibration.

VERDICT r04 weak 3: the network-level gradient claim of the DEFAULT assignment (bf16x3 head + fp16 tail) had no test.  A randomly initialised
50-layer ReLU network under train-mode BatchNorm is a chaotic map -- the stock fp32 layers themselves are 2e-2 off fp64 in the gradients -- so
the gradient gates are an all-parameter relative error + cosine and a per-stage cosine (a broken layer drives the cosine of its own and of
every earlier stage towards 0), not 1e-3; the forward quantities carry north_star's 1e-3."""
import copy
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))
pytestmark = pytest.mark.gpu


def rel(a, c):
    a, c = a.double(), c.double()
    return ((a - c).norm() / c.norm().clamp_min(1e-30)).item()


# (mode, gates: embeds, per-frame logits, all-gradient rel-L2, all-gradient cosine, worst per-stage cosine).  Measured in round 4
# (profiles/r04_e1_parity.txt): default 2.1e-4 / 6.0e-4 / ~0.12 (cosine ~0.99); bf16x3 1.1e-4 / 3.2e-4 / 0.098 (0.995); stock fp32: 6e-6 / 2e-5 / 0.023
# bn = 'tied' (round 6): the WELL-CONDITIONED full-depth check -- the same train-mode forward and backward, with the fp64 stock
# layers evaluated on the HIP path's OWN branch decisions (every ReLU pattern read back from the operand planes the encoder saved, the stem
# max-pool's argmax from its 1-byte index map: oracle/backbones_ref.REPLAY), as the generator / critic / VGG tests do.  Both sides then are the same
# piecewise-linear map, ReLU / argmax ties no longer dominate (the plain comparison has the stock fp32 layers 2e-2 from fp64 -- and an eval-mode
# BatchNorm run measured the same 0.11 - 0.15, so the spread is the ties on noise frames, not the batch coupling), and the all-parameter gradient
# gate can be tight: <= 1e-2.  (gates: embeds, logits, all-gradient rel-L2, cosine, worst per-stage cosine)
@pytest.mark.parametrize('mode,bn,gates', [('default', 'train', (5e-4, 1.5e-3, 0.2, 0.985, 0.95)), ('bf16x3', 'train', (3e-4, 8e-4, 0.16, 0.99, 0.95)),
                                          ('default', 'tied', (5e-4, 1.5e-3, 1e-2, 0.9999, 0.9999)), ('bf16x3', 'tied', (3e-4, 8e-4, 2e-3, 0.99999, 0.99999))])
def test_identity_encoder_full_geometry_forward_and_gradients(monkeypatch, mode, bn, gates):
    for k in ('LP_PREC', 'LP_PREC_E', 'LP_E_F16_TAIL', 'LP_E_HEAD_F16'):
        monkeypatch.delenv(k, raising=False)
    if mode != 'default':
        monkeypatch.setenv('LP_PREC', mode)
    from embedders import backbones
    from dataloaders.synthetic_voxceleb2 import make_sample
    from oracle import backbones_ref as BR
    torch.manual_seed(123)
    net = backbones.resnext50_32x4d(num_classes=512).cuda().train()
    ref = copy.deepcopy(net).double()
    b, k, size = 8, 8, 256
    x = torch.stack([make_sample(i, size, k, 98000, False, 123)[0]['enc_rgbs'] for i in range(b)]).cuda().reshape(b * k, 3, size, size)
    m32 = copy.deepcopy(net)
    r = torch.randn(b, 512, device='cuda')
    y = net(x)
    assert net.__dict__.get('_hip_param_names') is not None, 'the HIP path did not run'
    emb = y.view(b, k, -1).mean(1)
    replay = None
    if bn == 'tied':          # the branch decisions of the HIP forward (saved state of its autograd node; released by its backward)
        fn = y.grad_fn
        nchw = lambda t: t.permute(0, 3, 1, 2)
        _, y0, st0, idx, _ = fn.stem
        relu = [nchw(y0 * st0.scale + st0.shift > 0)]
        for sv in fn.blocks:
            c1, c2, c3 = sv[3].c, sv[6].c, sv[12].c
            relu += [nchw(sv[3].hi[..., :c1] > 0), nchw(sv[6].hi[..., :c2] > 0), nchw(sv[12].hi[..., :c3] > 0)]
        replay = {'relu': relu, 'pool': nchw(idx).long()}
    (emb * r).sum().backward()

    def stock(model, xin):          # the stock layers, on the recorded branch decisions when tie-masked
        BR.REPLAY = None if replay is None else {'relu': list(replay['relu']), 'pool': replay['pool']}
        try:
            out = BR.resnext_forward(model, xin)
        finally:
            left = 0 if BR.REPLAY is None else len(BR.REPLAY['relu'])
            BR.REPLAY = None
        assert left == 0, f'{left} recorded ReLU sites were not consumed by the oracle'
        return out
    yr = stock(ref, x.double())
    embr = yr.view(b, k, -1).mean(1)
    (embr * r.double()).sum().backward()
    y32 = stock(m32, x)
    e32 = y32.view(b, k, -1).mean(1)
    (e32 * r).sum().backward()
    torch.cuda.synchronize()
    names = [n for n, _ in net.named_parameters()]
    g, gr, g32 = [p.grad.double() for p in net.parameters()], [q.grad for q in ref.parameters()], [p.grad.double() for p in m32.parameters()]

    def grel(a, c):
        return float((sum(((u - v) ** 2).sum() for u, v in zip(a, c)) / sum((v ** 2).sum() for v in c)).sqrt())

    def gcos(a, c):
        fa, fc = torch.cat([t.reshape(-1) for t in a]), torch.cat([t.reshape(-1) for t in c])
        return float((fa * fc).sum() / (fa.norm() * fc.norm()))
    stages = {}
    for st in ('conv1|bn1', 'layer1', 'layer2', 'layer3', 'layer4', 'fc'):
        idx = [i for i, n in enumerate(names) if any(n.startswith(p_) for p_ in st.split('|'))]
        stages[st] = {'cosine': gcos([g[i] for i in idx], [gr[i] for i in idx]), 'rel': grel([g[i] for i in idx], [gr[i] for i in idx]),
                      'stock_fp32_rel': grel([g32[i] for i in idx], [gr[i] for i in idx])}
    modes = [{0: 'bf16', 1: 'bf16x3', 2: 'f16'}[m] for m in net.block_precs()]
    res = {'geometry': f'64 frames (8 samples x 8) of 256 x 256, train-mode BatchNorm' + (', fp64 stock layers on the HIP path\'s ReLU / max-pool branch decisions (tie-masked)' if bn == 'tied' else '') + ', U[0,1) frames', 'blocks': modes,
           'embeds': rel(emb, embr), 'per_frame_logits': rel(y, yr), 'all_gradients_rel': grel(g, gr), 'all_gradients_cosine': gcos(g, gr),
           'per_stage': stages,
           'stock_fp32_layers_vs_fp64': {'embeds': rel(e32, embr), 'per_frame_logits': rel(y32, yr), 'all_gradients_rel': grel(g32, gr),
                                         'all_gradients_cosine': gcos(g32, gr)}}
    print(f'[e1-full] mode {mode}, {"tie-masked" if bn == "tied" else "plain"} ({modes.count("f16")} fp16 blocks of {len(modes)}):', json.dumps(res))
    keep = os.environ.get('LP_PARITY_OUT')
    if keep:
        import bench
        path = os.path.join(keep, f'{bench.ROUND}_parity_gradients_{"f16" if mode == "default" else mode}.json')
        try:
            cur = json.load(open(path))
        except Exception:
            cur = {}
        cur['identity_encoder' if bn == 'train' else 'identity_encoder_tie_masked'] = res
        cur['stamp'] = bench.source_stamp()
        json.dump(cur, open(path, 'w'), indent=1)
    assert all(torch.isfinite(t).all() for t in g)
    worst_stage = min(v['cosine'] for v in stages.values())
    assert res['embeds'] < gates[0] and res['per_frame_logits'] < gates[1], res
    assert res['all_gradients_rel'] < gates[2] and res['all_gradients_cosine'] > gates[3] and worst_stage > gates[4], res
