"""Pose encoder (torchvision mobilenet_v2(num_classes=256), embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:28,56-58) at
the configs[2] geometry -- 8 frames of 256 x 256, train-mode BatchNorm -- forward AND backward through the HIP encoder (bf16x3 contractions) against the
stock layers of oracle/backbones_ref.py in fp64, with the stock fp32 layers against the same fp64 run as the calibration.

A randomly initialised 52-layer ReLU6 network under train-mode BatchNorm over 8 frames is a chaotic map (the stock fp32 layers are 3e-2 off fp64 in the
gradients: tests/test_mobilenet_train_hip.py), so -- as for the generator, the critic, the VGG stacks and the identity encoder -- the WELL-CONDITIONED
full-depth check evaluates the fp64 stock layers on the HIP path's OWN branch decisions: each of the 35 ReLU6 sites takes the (linear, saturated) pattern
the HIP forward took (read back from the tensors its autograd node saved: oracle/backbones_ref.REPLAY['relu6']).  Both sides then are the same
piecewise-linear map and what remains is arithmetic: the all-parameter gradient is gated at 1e-3 (SURVEY 8d's own figure)."""
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


def _pattern(y, st):
    """(linear, saturated) NCHW masks of relu6(y * scale + shift) as the kernels take them: one rounding of the affine (fmaf), then the two thresholds"""
    v = (y.double() * st.scale.double() + st.shift.double()).float()
    lin, sat = (v > 0) & (v < 6), v >= 6
    return lin.permute(0, 3, 1, 2), sat.permute(0, 3, 1, 2)


@pytest.mark.parametrize('tied,gates', [(False, (2e-4, 0.5, 0.9)), (True, (2e-4, 1e-3, 0.999999))])
def test_pose_encoder_full_geometry_forward_and_gradients(tied, gates):
    """gates: pose vector, all-gradient rel-L2, all-gradient cosine"""
    from embedders import backbones
    from dataloaders.synthetic_voxceleb2 import make_sample
    from oracle import backbones_ref as BR
    torch.manual_seed(321)
    net = backbones.mobilenet_v2(256).cuda().train()
    net.classifier[0].p = 0.0                      # Dropout off on both sides (its random mask is not part of parity)
    ref, m32 = copy.deepcopy(net).double(), copy.deepcopy(net)
    b, size = 8, 256
    x = torch.stack([make_sample(i, size, 8, 98000, False, 321)[0]['pose_input_rgbs'] for i in range(b)]).cuda().reshape(b, 3, size, size)
    r = torch.randn(b, 256, device='cuda')
    y = net(x)
    assert net.__dict__.get('_hip_feature_param_names') is not None, 'the HIP training path did not run'
    replay = None
    if tied:          # the branch decisions of the HIP forward (saved state of the features Function; released by its backward)
        todo, fn = [y.grad_fn], None
        while todo and fn is None:          # (the features Function sits behind the classifier and the Dropout node)
            node = todo.pop(0)
            if hasattr(node, 'stem') and hasattr(node, 'saved') and hasattr(node, 'last'):
                fn = node
            else:
                todo += [f for f, _ in node.next_functions if f is not None]
        assert fn is not None, 'MobileNetFeaturesFunction node not found behind the classifier'
        _, y0, st0, _ = fn.stem
        sites = [_pattern(y0, st0)]
        for rec in fn.saved:
            if rec['expand']:
                sites.append(_pattern(rec['raw'], rec['st_raw']))
            sites.append(_pattern(rec['yd'], rec['std']))
        _, yl, stl, _, _ = fn.last
        sites.append(_pattern(yl, stl))
        replay = sites
    (y * r).sum().backward()

    def stock(model, xin):
        BR.REPLAY = None if replay is None else {'relu6': list(replay)}
        try:
            out = BR.mobilenet_forward(model, xin)
        finally:
            left = 0 if BR.REPLAY is None else len(BR.REPLAY['relu6'])
            BR.REPLAY = None
        assert left == 0, f'{left} recorded ReLU6 sites were not consumed by the oracle'
        return out
    yr = stock(ref, x.double())
    (yr * r.double()).sum().backward()
    y32 = stock(m32, x)
    (y32 * r).sum().backward()
    torch.cuda.synchronize()
    g, gr, g32 = [p.grad.double() for p in net.parameters()], [q.grad for q in ref.parameters()], [p.grad.double() for p in m32.parameters()]

    def grel(a, c):
        return float((sum(((u - v) ** 2).sum() for u, v in zip(a, c)) / sum((v ** 2).sum() for v in c)).sqrt())

    def gcos(a, c):
        fa, fc = torch.cat([t.reshape(-1) for t in a]), torch.cat([t.reshape(-1) for t in c])
        return float((fa * fc).sum() / (fa.norm() * fc.norm()))
    # (per tensor; a BatchNorm bias in front of conv -> train-mode BatchNorm has an exactly zero gradient -- the statistics remove any per-channel shift --
    #  so tensors whose reference gradient is below 1e-6 of the largest one are rounding residue on both sides and are left out of this column)
    top = max(float(v.norm()) for v in gr)
    worst = max(((n, rel(u, v)) for (n, _), u, v in zip(net.named_parameters(), g, gr) if float(v.norm()) > 1e-6 * top), key=lambda kv: kv[1])
    res = {'geometry': '8 frames of 256 x 256, train-mode BatchNorm, bf16x3 contractions' + (', fp64 stock layers on the HIP path\'s ReLU6 branch decisions (tie-masked)' if tied else ''),
           'relu6_sites': None if replay is None else len(replay),
           'pose_vector': rel(y, yr), 'all_gradients_rel': grel(g, gr), 'all_gradients_cosine': gcos(g, gr), 'worst_parameter_tensor': list(worst),
           'stock_fp32_layers_vs_fp64': {'pose_vector': rel(y32, yr), 'all_gradients_rel': grel(g32, gr), 'all_gradients_cosine': gcos(g32, gr)}}
    print(f'[e2-full] {"tie-masked" if tied else "plain"}:', json.dumps(res))
    keep = os.environ.get('LP_PARITY_OUT')
    if keep:
        import bench
        for mode in ('f16', 'bf16x3'):          # (the pose encoder runs bf16x3 contractions in every assignment: the same figure goes into both files)
            path = os.path.join(keep, f'{bench.ROUND}_parity_gradients_{mode}.json')
            try:
                cur = json.load(open(path))
            except Exception:
                cur = {}
            cur['pose_encoder_tie_masked' if tied else 'pose_encoder'] = res
            cur['stamp'] = bench.source_stamp()
            json.dump(cur, open(path, 'w'), indent=1)
    assert all(torch.isfinite(t).all() for t in g)
    assert res['pose_vector'] < gates[0], res
    assert res['all_gradients_rel'] < gates[1] and res['all_gradients_cosine'] > gates[2], res
