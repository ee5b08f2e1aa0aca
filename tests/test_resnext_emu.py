"""CPU check of the ORCHESTRATION of the embedder's HIP path (embedders/resnext_hip.py): with every ``hipops`` entry point replaced by
its plain-torch emulation (tests/emu_ops.py), the ResNeXt-50 autograd.Function -- which kernel runs on which tensor in which order,
what is saved for backward, how stride-2 blocks / downsample branches / the stem / the classifier are wired -- must reproduce the stock-layer
evaluation (oracle/backbones_ref.py) -- output, every parameter gradient and every BatchNorm buffer update, in train and in eval mode (fp64: to rounding).
The kernels themselves are checked against the same emulation functions on the GPU (tests/test_resnext_hip.py)."""
import copy
import os
import sys

import pytest
import torch

from oracle import backbones_ref as BR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize('mode', ['f16', 'bf16x3'])         # f16: conv outputs 16-bit resident (the y16 wiring); bf16x3: fp32 outputs
@pytest.mark.parametrize('train', [True, False])
def test_resnext_function_matches_stock_autograd(monkeypatch, train, mode):
    import emu_ops
    monkeypatch.setenv('LP_PREC_E', mode)
    from embedders import resnext_hip
    from embedders.backbones import resnext50_32x4d
    from latent_pose_reenactment_amd import hipops
    monkeypatch.setattr(resnext_hip, 'ops', emu_ops)
    monkeypatch.setattr(hipops, 'PackBatch', emu_ops.PackBatch)
    monkeypatch.setattr(hipops, 'pack_grouped', emu_ops.pack_grouped)
    torch.manual_seed(3)
    m = resnext50_32x4d(16).double()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.uniform_(-0.3, 0.3)
            mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 1.5)
    m2 = copy.deepcopy(m)
    m.train(train); m2.train(train)
    x = torch.rand(8, 3, 64, 64, dtype=torch.double)      # (the emulation has no geometry limits: small = fast)
    r = torch.randn(8, 16, dtype=torch.double)
    y_ref = BR.resnext_forward(m, x)
    (y_ref * r).sum().backward()
    assert resnext_hip.supported(8, 128, 128) and not resnext_hip.supported(8, 64, 64) and not resnext_hip.supported(2, 128, 128)
    m2._hip_structure()
    y = resnext_hip.ResNeXtFunction.apply(m2, x, *[p for _, p in m2.named_parameters()])
    (y * r).sum().backward()
    assert rel(y, y_ref) < 1e-10
    for (k, p1), (_, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert p2.grad is not None and rel(p2.grad, p1.grad) < 1e-9, k
    for (k, b1), (_, b2) in zip(m.named_buffers(), m2.named_buffers()):
        assert rel(b2, b1) < 1e-10, k


@pytest.mark.parametrize('train', [True, False])
def test_mobilenet_training_path_matches_stock_autograd(monkeypatch, train):
    """MobileNetV2 with autograd on (embedders/mobilenet_hip.py): features Function + Dropout + classifier Function vs the stock
    module.  (The bias of a block's last BatchNorm feeds a linear conv followed by a train-mode BatchNorm, so its true gradient is 0:
    errors are measured against the global gradient scale.)"""
    import emu_ops
    from embedders import mobilenet_hip, resnext_hip
    from embedders.backbones import mobilenet_v2
    from latent_pose_reenactment_amd import hipops
    monkeypatch.setattr(resnext_hip, 'ops', emu_ops)
    monkeypatch.setattr(mobilenet_hip, 'ops', emu_ops)
    monkeypatch.setattr(hipops, 'PackBatch', emu_ops.PackBatch)
    monkeypatch.setattr(emu_ops, 'pack_weights', lambda w, mode, prec, small_k=False: emu_ops.Pack(w, mode), raising=False)
    torch.manual_seed(5)
    m = mobilenet_v2(16).double()
    m.classifier[0].p = 0.0
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.uniform_(-0.3, 0.3)
            mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 1.5)
    m2 = copy.deepcopy(m)
    m.train(train); m2.train(train)
    x = torch.rand(8, 3, 64, 64, dtype=torch.double)
    r = torch.randn(8, 16, dtype=torch.double)
    y_ref = BR.mobilenet_forward(m, x)
    (y_ref * r).sum().backward()
    y = m2._forward_hip_train(x)
    (y * r).sum().backward()
    assert rel(y, y_ref) < 1e-10
    scale = max(p.grad.norm().item() for p in m.parameters())
    for (k, p1), (_, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert p2.grad is not None and (p2.grad - p1.grad).norm().item() < 1e-9 * scale, k
    for (k, b1), (_, b2) in zip(m.named_buffers(), m2.named_buffers()):
        assert rel(b2, b1) < 1e-10, k
