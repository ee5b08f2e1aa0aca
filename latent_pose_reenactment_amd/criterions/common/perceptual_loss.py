"""VGG perceptual loss (reference: criterions/common/perceptual_loss.py:19-110).

``net='caffe'``: Caffe VGG-19, ``net='face'``: VGGFace (VGG-16); first 30 ``features`` modules with MaxPool replaced by
AvgPool; inputs mapped (x+1)/2 then (x - mean_bgr/255)*255 on RGB-ordered channels (kept as in the reference); the loss
is the sum over all 13 ReLU outputs of mean|f(fake) - f(real)| times ``weight``.  The VGG definitions come from
torchvision in the reference (absent here); the standard 'E' / 'D' configurations are restated below.  State-dict keys of
``self.model`` are the ``features`` indices (``0.weight`` ...), as in the reference.
Round-1 status: stock PyTorch-ROCm convolutions (generator-only HIP scope); maps onto lp_conv_fwd next."""
import os
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn

CFG = {
    'vgg19': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M'],
    'vgg16': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M'],
}
WEIGHT_FILES = {'caffe': ('vgg19', 'vgg19-d01eb7cb.pth'), 'face': ('vgg16', 'vgg_face_weights.pth')}


def build_features(cfg, num_layers=30, width_div=1):
    layers, cin = [], 3
    for v in cfg:
        if v == 'M':
            layers.append(nn.AvgPool2d(kernel_size=2, stride=2, padding=0))
        else:
            layers += [nn.Conv2d(cin, v // width_div, 3, padding=1), nn.ReLU(inplace=False)]
            cin = v // width_div
    return nn.Sequential(*layers[:num_layers])


class PerceptualLoss(nn.Module):
    def __init__(self, weight, vgg_weights_dir, net='caffe', synthetic_seed=None, width_div=1):
        """``synthetic_seed`` (not in the reference): when the external weight file is absent (no network in the build
        image) initialise He-normal from that seed instead of failing -- used by bench.py / tests only."""
        super().__init__()
        if net not in WEIGHT_FILES:
            raise ValueError(f"Unknown type of PerceptualLoss: expected '{{caffe,face}}', got '{net}'")
        arch, fname = WEIGHT_FILES[net]
        self.weight = weight
        self.model = build_features(CFG[arch], 30, width_div)
        path = os.path.join(str(vgg_weights_dir), fname)
        if os.path.exists(path):
            sd = torch.load(path, map_location='cpu')
            sd = OrderedDict((k[len('features.'):] if k.startswith('features.') else k, v) for k, v in sd.items()
                             if not k.startswith('classifier'))
            self.model.load_state_dict(sd)
        elif synthetic_seed is not None:
            g = torch.Generator().manual_seed(synthetic_seed)
            for m in self.model:
                if isinstance(m, nn.Conv2d):
                    fan_in = m.weight[0].numel()
                    with torch.no_grad():
                        m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                        m.bias.zero_()
        else:
            raise FileNotFoundError(f'{path} not found (download per the reference INSTALL.md)')
        for p in self.model.parameters():
            p.requires_grad = False
        mean = torch.tensor([103.939, 116.779, 123.680]) / 255.
        std = torch.tensor([1., 1., 1.]) / 255.
        self.register_buffer('mean', mean[None, :, None, None])
        self.register_buffer('std', std[None, :, None, None])

    def normalize_inputs(self, x):
        return (x - self.mean) / self.std

    def forward(self, input, target):
        fi = self.normalize_inputs((input + 1) / 2)
        ft = self.normalize_inputs((target.detach() + 1) / 2)
        loss = 0
        for layer in self.model:
            fi, ft = layer(fi), layer(ft)
            if isinstance(layer, nn.ReLU):
                loss = loss + F.l1_loss(fi, ft)
        return loss * self.weight
