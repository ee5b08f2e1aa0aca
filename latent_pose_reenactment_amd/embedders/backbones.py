"""ResNeXt-50 32x4d and MobileNetV2 with torchvision-0.6-compatible ``state_dict`` keys.

The reference instantiates both from torchvision (embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:26-28),
which is an un-vendored dependency and absent from this image; these are restatements of the public architectures
(He et al. / Xie et al. ResNeXt; Sandler et al. MobileNetV2) so that reference checkpoints load by key.  They run on
stock PyTorch-ROCm ops: the embedder is the LAST row of the hot-path plan, not yet hand-written HIP."""
import torch
import torch.nn.functional as F
from torch import nn

_PENDING_COUNTERS = []


class _BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters, buffers and state_dict keys) whose ``num_batches_tracked += 1`` -- one tiny launch per
    layer per forward in train mode (52 per MobileNetV2 pass) -- is deferred and issued as ONE multi-tensor add by the backbone
    at the end of its forward (momentum is a constant here, so the counter does not enter the statistics update)."""

    def forward(self, x):
        if self.training and self.track_running_stats:
            _PENDING_COUNTERS.append(self.num_batches_tracked)
            return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, True, self.momentum, self.eps)
        return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, not self.track_running_stats, 0.0, self.eps)


def _flush_bn_counters():
    if _PENDING_COUNTERS:
        torch._foreach_add_(_PENDING_COUNTERS, 1)
        _PENDING_COUNTERS.clear()


# ---------------------------------------------------------------- ResNeXt
class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample, groups, base_width):
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = _BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, groups=groups, bias=False)
        self.bn2 = _BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = _BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


class ResNeXt(nn.Module):
    def __init__(self, layers, groups, width_per_group, num_classes):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = _BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._stage(64, layers[0], 1, groups, width_per_group)
        self.layer2 = self._stage(128, layers[1], 2, groups, width_per_group)
        self.layer3 = self._stage(256, layers[2], 2, groups, width_per_group)
        self.layer4 = self._stage(512, layers[3], 2, groups, width_per_group)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def _stage(self, planes, blocks, stride, groups, base_width):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), _BatchNorm2d(planes * 4))
        seq = [_Bottleneck(self.inplanes, planes, stride, down, groups, base_width)]
        self.inplanes = planes * 4
        seq += [_Bottleneck(self.inplanes, planes, 1, None, groups, base_width) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        _flush_bn_counters()
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnext50_32x4d(num_classes=1000):
    return ResNeXt([3, 4, 6, 3], 32, 4, num_classes)


# ---------------------------------------------------------------- MobileNetV2
class _ConvBNReLU(nn.Sequential):
    def __init__(self, cin, cout, k=3, stride=1, groups=1):
        super().__init__(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), _BatchNorm2d(cout),
                         nn.ReLU6(inplace=True))


class _InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride, t):
        super().__init__()
        hidden = int(round(inp * t))
        self.use_res = stride == 1 and inp == oup
        layers = []
        if t != 1:
            layers.append(_ConvBNReLU(inp, hidden, 1))
        layers += [_ConvBNReLU(hidden, hidden, 3, stride, groups=hidden), nn.Conv2d(hidden, oup, 1, 1, 0, bias=False),
                   _BatchNorm2d(oup)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.use_res else self.conv(x)


class MobileNetV2(nn.Module):
    CFG = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]]

    def __init__(self, num_classes=1000):
        super().__init__()
        feats = [_ConvBNReLU(3, 32, stride=2)]
        cin = 32
        for t, c, n, s in self.CFG:
            for i in range(n):
                feats.append(_InvertedResidual(cin, c, s if i == 0 else 1, t))
                cin = c
        feats.append(_ConvBNReLU(cin, 1280, 1))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(1280, num_classes))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out')
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.features(x)
        _flush_bn_counters()
        return self.classifier(x.mean([2, 3]))


def mobilenet_v2(num_classes=1000):
    return MobileNetV2(num_classes)
