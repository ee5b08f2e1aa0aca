"""TEST INFRASTRUCTURE -- the stock-layer forward of the two encoder backbones (oracle for SURVEY rows E1 / E2).

The reference instantiates its identity / pose encoders from torchvision 0.6.1 (embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:
26-28: ``resnext50_32x4d(num_classes=512)``, ``mobilenet_v2(num_classes=256)``), an un-vendored dependency that is absent from this image:
PARITY UNPINNED for the architecture itself -- what is restated here are the public definitions (Xie et al. ResNeXt / torchvision
``Bottleneck``; Sandler et al. MobileNetV2 / torchvision ``InvertedResidual``) evaluated with stock ``torch.nn.functional`` ops
(``conv2d``, ``batch_norm``, ``max_pool2d``, ``linear``: MIOpen / rocBLAS on the GPU, oneDNN on the CPU) over the PARAMETERS AND BUFFERS of the
product's container modules (``latent_pose_reenactment_amd/embedders/backbones.py``: torchvision-compatible ``state_dict`` keys, no arithmetic).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.  The product's encoders run on the
hand-written gfx950 kernels or raise; they contain no stock-layer forward.  ``stock_layers()`` is how a TEST runs a product module tree
(TrainingModule, the embedder plugin) on the stock layers: it swaps the ``forward`` of the two backbone classes for the functions below
while the context is active (geometries the HIP encoders do not cover: 32-px golden fixtures; A/B regression runs; the CPU baseline)."""
import contextlib

import torch
import torch.nn.functional as F


def _bn(m, x):
    """nn.BatchNorm2d.forward on the container's parameters / buffers (momentum is a constant in both backbones)"""
    if m.training and m.track_running_stats:
        m.num_batches_tracked += 1
        return F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, True, m.momentum, m.eps)
    return F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, not m.track_running_stats, 0.0, m.eps)


def _conv(m, x):
    return F.conv2d(x, m.weight, m.bias, m.stride, m.padding, m.dilation, m.groups)


# Tie-masked evaluation of the ResNeXt oracle (round 6, tests/test_e1_full_gpu.py): a pre-activation within rounding distance of 0 flips its ReLU --
# and a 3x3 max-pool window with two near-equal candidates its argmax -- between two correct implementations, and on a 50-layer network those
# flips, not arithmetic error, dominate any plain gradient comparison (the stock fp32 layers are 2e-2 from fp64).  When REPLAY is a dict
# {'relu': [bool NCHW masks in execution order], 'pool': int64 NCHW argmax (ky * 3 + kx) of the stem's max-pool}, the oracle takes the OTHER
# implementation's branch decisions: relu(x) := x * mask, maxpool := gather at the recorded window position -- the same piecewise-linear map on
# both sides, so what remains is arithmetic.
REPLAY = None


def _relu(x):
    if REPLAY is None:
        return F.relu(x)
    m = REPLAY['relu'].pop(0)
    assert m.shape == x.shape, (m.shape, x.shape)
    return x * m.to(x.dtype)


def _maxpool(x):
    if REPLAY is None:
        return F.max_pool2d(x, 3, 2, 1)
    idx = REPLAY['pool']
    n, c, h, w = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    win = F.unfold(x, 3, padding=1, stride=2).view(n, c, 9, ho, wo)          # window position k = ky * 3 + kx
    return win.gather(2, idx.view(n, c, 1, ho, wo)).squeeze(2)


# ---------------------------------------------------------------- ResNeXt-50 32x4d (torchvision resnet.py: ResNet / Bottleneck)
def _bottleneck(blk, x):
    idt = x
    if blk.downsample is not None:
        idt = _bn(blk.downsample[1], _conv(blk.downsample[0], x))
    out = _relu(_bn(blk.bn1, _conv(blk.conv1, x)))
    out = _relu(_bn(blk.bn2, _conv(blk.conv2, out)))
    out = _bn(blk.bn3, _conv(blk.conv3, out))
    return _relu(out + idt)


def resnext_forward(net, x):
    """frames [N,3,H,W] -> logits [N, num_classes]"""
    x = _maxpool(_relu(_bn(net.bn1, _conv(net.conv1, x))))
    for stage in (net.layer1, net.layer2, net.layer3, net.layer4):
        for blk in stage:
            x = _bottleneck(blk, x)
    return F.linear(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1), net.fc.weight, net.fc.bias)


# ---------------------------------------------------------------- MobileNetV2 (torchvision mobilenet.py: ConvBNReLU / InvertedResidual)
def _relu6(x):
    """ReLU6, or -- tie-masked evaluation (REPLAY['relu6'] = [(linear, saturated) bool NCHW masks in execution order], tests/test_e2_full_gpu.py) --
    the other implementation's branch decisions: x where it took the linear branch, 6 where it saturated, 0 elsewhere"""
    if REPLAY is None or 'relu6' not in REPLAY:
        return F.relu6(x)
    lin, sat = REPLAY['relu6'].pop(0)
    assert lin.shape == x.shape, (lin.shape, x.shape)
    return x * lin.to(x.dtype) + 6.0 * sat.to(x.dtype)


def _conv_bn_relu6(seq, x):
    return _relu6(_bn(seq[1], _conv(seq[0], x)))


def _inverted_residual(blk, x):
    layers = list(blk.conv)
    h = x
    for m in layers[:-2]:
        h = _conv_bn_relu6(m, h)
    h = _bn(layers[-1], _conv(layers[-2], h))
    return x + h if blk.use_res else h


def mobilenet_forward(net, x):
    feats = list(net.features)
    x = _conv_bn_relu6(feats[0], x)
    for blk in feats[1:-1]:
        x = _inverted_residual(blk, x)
    x = _conv_bn_relu6(feats[-1], x)
    drop, fc = net.classifier[0], net.classifier[1]
    return F.linear(F.dropout(x.mean([2, 3]), drop.p, drop.training), fc.weight, fc.bias)


def forward(net, x):
    return resnext_forward(net, x) if hasattr(net, 'layer1') else mobilenet_forward(net, x)


@contextlib.contextmanager
def stock_layers():
    """while active, EVERY instance of the product's two backbone classes evaluates on the stock layers above (a class-level swap of
    ``forward``: module trees that were deep-copied -- the EMA copies of TrainingModule -- follow it too)"""
    from latent_pose_reenactment_amd.embedders import backbones
    mods = [backbones]
    try:                     # the plugin loader imports the same file as top-level package ``embedders`` (reference layout): patch both
        import embedders.backbones as plugin_backbones
        if plugin_backbones is not backbones:
            mods.append(plugin_backbones)
    except ImportError:
        pass
    saved = [(m.ResNeXt, m.ResNeXt.forward, m.MobileNetV2, m.MobileNetV2.forward) for m in mods]
    for m in mods:
        m.ResNeXt.forward = resnext_forward
        m.MobileNetV2.forward = mobilenet_forward
    try:
        yield
    finally:
        for rx, rf, mb, mf in saved:
            rx.forward, mb.forward = rf, mf
