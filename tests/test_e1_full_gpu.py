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
# bn = 'eval' (round 6): the WELL-CONDITIONED full-depth check.  The running statistics are first calibrated to this very batch
# (one train-mode forward of the fp64 stock layers with momentum 1), then all three nets run in eval mode: the same 53 convolutions, BatchNorms,
# ReLUs and the same activation scale as the training forward, but every BatchNorm is a constant per-channel affine map -- no batch coupling, no
# chaos -- so an arithmetic error in ANY layer shows up undiminished and the gradient gate can be tight: all-parameter rel-L2 <= 1e-2
# (gates: embeds, logits, all-gradient rel-L2, cosine, worst per-stage cosine).
@pytest.mark.parametrize('mode,bn,gates', [('default', 'train', (5e-4, 1.5e-3, 0.2, 0.985, 0.95)), ('bf16x3', 'train', (3e-4, 8e-4, 0.16, 0.99, 0.95)),
                                          ('default', 'eval', (5e-4, 1e-3, 1e-2, 0.9999, 0.9999)), ('bf16x3', 'eval', (1e-4, 1e-4, 2e-3, 0.99999, 0.99999))])
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
    if bn == 'eval':
        for mod in ref.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.momentum = 1.0                      # running statistics := the statistics of this batch
        with torch.no_grad():
            BR.resnext_forward(ref, x.double())
        net.load_state_dict({k_: v.float() if v.dtype.is_floating_point else v for k_, v in ref.state_dict().items()})
        net.eval(); ref.eval()
    m32 = copy.deepcopy(net)
    r = torch.randn(b, 512, device='cuda')
    y = net(x)
    assert net.__dict__.get('_hip_param_names') is not None, 'the HIP path did not run'
    emb = y.view(b, k, -1).mean(1)
    (emb * r).sum().backward()
    yr = BR.resnext_forward(ref, x.double())
    embr = yr.view(b, k, -1).mean(1)
    (embr * r.double()).sum().backward()
    y32 = BR.resnext_forward(m32, x)
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
    res = {'geometry': f'64 frames (8 samples x 8) of 256 x 256, {bn}-mode BatchNorm' + (' (running statistics calibrated to the batch)' if bn == 'eval' else '') + ', U[0,1) frames', 'blocks': modes,
           'embeds': rel(emb, embr), 'per_frame_logits': rel(y, yr), 'all_gradients_rel': grel(g, gr), 'all_gradients_cosine': gcos(g, gr),
           'per_stage': stages,
           'stock_fp32_layers_vs_fp64': {'embeds': rel(e32, embr), 'per_frame_logits': rel(y32, yr), 'all_gradients_rel': grel(g32, gr),
                                         'all_gradients_cosine': gcos(g32, gr)}}
    print(f'[e1-full] mode {mode}, {bn}-mode BatchNorm ({modes.count("f16")} fp16 blocks of {len(modes)}):', json.dumps(res))
    keep = os.environ.get('LP_PARITY_OUT')
    if keep:
        import bench
        path = os.path.join(keep, f'{bench.ROUND}_parity_gradients_{"f16" if mode == "default" else mode}.json')
        try:
            cur = json.load(open(path))
        except Exception:
            cur = {}
        cur['identity_encoder' if bn == 'train' else 'identity_encoder_eval_bn'] = res
        cur['stamp'] = bench.source_stamp()
        json.dump(cur, open(path, 'w'), indent=1)
    assert all(torch.isfinite(t).all() for t in g)
    worst_stage = min(v['cosine'] for v in stages.values())
    assert res['embeds'] < gates[0] and res['per_frame_logits'] < gates[1], res
    assert res['all_gradients_rel'] < gates[2] and res['all_gradients_cosine'] > gates[3] and worst_stage > gates[4], res
