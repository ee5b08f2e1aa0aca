"""VGG perceptual loss (reference: criterions/common/perceptual_loss.py:19-110).

``net='caffe'``: Caffe VGG-19, ``net='face'``: VGGFace (VGG-16); first 30 ``features`` modules with MaxPool replaced by
AvgPool; inputs mapped (x+1)/2 then (x - mean_bgr/255)*255 on RGB-ordered channels (kept as in the reference); the loss
is the sum over all 13 ReLU outputs of mean|f(fake) - f(real)| times ``weight``.  The VGG definitions come from
torchvision in the reference (absent here); the standard 'E' / 'D' configurations are restated below.  State-dict keys of
``self.model`` are the ``features`` indices (``0.weight`` ...), as in the reference.
The feature stack runs on the gfx950 kernels: every conv is lp_conv16_fwd on 16-bit operand planes (relu fused into the producer's epilogue),
ReLU+AvgPool is lp_avgpool2_fwd, each tap is the fused L1-of-ReLUs kernel; the frozen weights are packed to bf16 once."""
import os
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn

from latent_pose_reenactment_amd import hipops as ops
from latent_pose_reenactment_amd import nn as lpnn
from latent_pose_reenactment_amd.nn import AvgPool2Fn, default_prec, hip_conv, hip_l1_tap, to_nhwc

CFG = {
    'vgg19': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M'],
    'vgg16': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M'],
}
FAKE16 = os.environ.get('LP_VGG_FAKE16', '1') != '0'        # ... and the generated image's pass planes-only as well (autograd through phantoms)
TAPS16 = os.environ.get('LP_VGG_TAPS16', '1') != '0'       # target-image taps as 16-bit planes in the fp16 mode (0: fp32 taps, the round-3 path)
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
            sd = OrderedDict((k, v) for k, v in sd.items() if int(k.split('.')[0]) < 30)     # only the first 30 modules are used
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

    def prep(self, x):
        """normalize_inputs((x + 1) / 2) of an image in [-1, 1] (perceptual_loss.py:86-93) as ONE launch that also leaves the NHWC layout the convs
        read (round 6; same fp32 operations in the same order) -> logical NCHW view of NHWC storage"""
        mean, std = self.__dict__.get('_ms', (None, None))
        if mean is None or mean.device != x.device:
            mean, std = self.mean.reshape(3).contiguous(), self.std.reshape(3).contiguous()
            self.__dict__['_ms'] = (mean, std)
        return lpnn.ImagePrepFn.apply(x, mean, std).permute(0, 3, 1, 2)

    def _packs(self, prec):
        """frozen weights: (forward, dgrad) bf16 packs per conv, built once per precision mode"""
        cache = self.__dict__.setdefault('_pack_cache', {})
        key = (prec, self.mean.device)
        if key not in cache:
            packs = {}
            for i, m in enumerate(self.model):
                if isinstance(m, nn.Conv2d):
                    w = m.weight.detach().contiguous()
                    packs[i] = (ops.pack_weights(w, 0, prec, small_k=w.shape[1] <= 32), ops.pack_weights(w, 1, prec, small_k=w.shape[0] <= 32))
            cache[key] = packs
        return cache[key]

    def _features(self, x, packs, prec, taps, targets=None):
        """``targets`` None: collect the taps (pre-ReLU conv outputs; the ReLU is fused into their consumers) into ``taps``.
        ``targets`` = taps of the other image: append the L1 term of every tap instead (the tap tensor flows on through
        L1TapFn so that its two gradients are summed inside the L1 backward kernel)."""
        if (targets is not None and targets and isinstance(targets[0], ops.Tap16) and prec == lpnn.PREC_F16 and FAKE16 and lpnn.RELU_TAPE is None
                and all(m.weight.shape[0] % 8 == 0 for m in self.model if isinstance(m, nn.Conv2d))):
            return self._features_planes(x, packs, prec, taps, targets)
        cur, pending_relu, cur16 = to_nhwc(x), False, None
        n_layers = len(self.model)
        for i, layer in enumerate(self.model):
            if isinstance(layer, nn.Conv2d):
                # conv -> ReLU -> conv: this conv's epilogue also writes the 16-bit operand planes of relu(y) for the next conv
                emit = 1 if (i + 2 < n_layers and isinstance(self.model[i + 2], nn.Conv2d)) else None
                out = hip_conv(cur, layer.weight, layer.bias, ksize=3, pro=2 if pending_relu else 0, prec=prec, packs=packs[i],
                               x16=cur16, emit16=emit)
                cur, cur16 = out if emit is not None else (out, None)
                pending_relu = True
            elif isinstance(layer, nn.ReLU):
                if cur16 is None:          # (a conv that emitted relu(y) planes has recorded this site already)
                    lpnn.tape_relu(lambda: cur > 0)
                if targets is None:
                    taps.append(cur)
                else:
                    cur, term = hip_l1_tap(cur, targets[len(taps)], relu_in=True)
                    taps.append(term)
            else:
                # pool -> conv: the pool launch also writes the next conv's operand planes (one-plane precision modes)
                holder = [] if (prec != lpnn.PREC_BF16X3 and i + 1 < n_layers and isinstance(self.model[i + 1], nn.Conv2d)) else None
                cur = AvgPool2Fn.apply(cur, pending_relu, None if holder is None else (prec, holder))
                cur16 = holder[0] if holder else None
                pending_relu = False
        return taps

    def _features_planes(self, x, packs, prec, taps, targets):
        """the generated image's pass WITH autograd and without fp32 activations (fp16 mode, 16-bit target taps): every conv writes only
        the operand planes of relu(y), the L1 taps and the pools read planes, and phantom tensors (nn.phantom: shape + autograd edge, no
        storage) connect the Functions; the gradients (fp32) are unchanged: sign pattern x scale from the L1 sites, the ReLU masks from
        the planes."""
        from latent_pose_reenactment_amd.nn import AvgPool2Fn16, hip_l1_tap16
        cur, cur16, first = to_nhwc(x), None, True
        for i, layer in enumerate(self.model):
            if isinstance(layer, nn.Conv2d):
                # (after a pool the planes hold the pooled tensor itself: no ReLU between pool and conv)
                cur, cur16 = hip_conv(cur, layer.weight, layer.bias, ksize=3, pro=2 if (not first and relu_pending) else 0, prec=prec, packs=packs[i],
                                      x16=cur16, emit16=1, want_y=False)
                first, relu_pending = False, True
            elif isinstance(layer, nn.ReLU):
                cur, term = hip_l1_tap16(cur, ops.Tap16(cur16, prec), targets[len(taps)])
                taps.append(term)
            else:
                holder = []
                cur = AvgPool2Fn16.apply(cur, cur16, prec, holder)
                cur16, relu_pending = holder[0], False
        return taps

    def _features16(self, x, packs, prec):
        """The taps of an image that needs NO gradient (the target image), without any fp32 activation: every conv writes only the 16-bit
        operand planes of relu(y) -- what the next conv consumes anyway -- the pools run plane to plane, and a tap IS those planes
        (``ops.Tap16``; the L1 kernel of the other image's pass decodes them).  One-plane fp16 mode only: the taps are then rounded to 11
        significant bits ONCE, on top of features that carry the operand rounding of up to 13 fp16 convs already (measured on the full step:
        tests/test_metatrain_full_gpu.py loss.VGG / loss.VGGFace).  Per 256 x 256 x 64 layer and image batch this avoids a 134 MB fp32
        store, 134 MB of L1 reads and 67 MB of pool reads."""
        cur16, taps = None, []
        x = to_nhwc(x)
        for i, layer in enumerate(self.model):
            if isinstance(layer, nn.Conv2d):
                bias = None if layer.bias is None else layer.bias.detach().contiguous()
                if cur16 is None:
                    n, h, w, cin = x.shape
                    if ops.thin_conv_supported(cin, layer.weight.shape[0], 3, w):
                        _, cur16 = ops.thin_conv(x, packs[i][0], ksize=3, bias=bias, prec=prec, out16=1, want_y=False)
                    else:
                        _, cur16 = ops.conv16(ops.act_pack(x, pro=0, prec=prec), packs[i][0], ksize=3, bias=bias, prec=prec, out16=1, want_y=False)
                else:
                    _, cur16 = ops.conv16(cur16, packs[i][0], ksize=3, bias=bias, prec=prec, out16=1, want_y=False)
            elif isinstance(layer, nn.ReLU):
                taps.append(ops.Tap16(cur16, prec))
            else:
                cur16 = ops.avgpool2_fwd16(cur16, prec)
        return taps

    def target_features(self, target):
        """the taps of the REAL image: no autograd, no dependence on the generator -- a training step may compute them early, on a
        side stream, beside the encoders and the generator (runners/holycow.py TrainingModule.forward, streams.py)"""
        if not target.is_cuda:
            raise RuntimeError('PerceptualLoss runs on the MI355X HIP path only (no CPU fallback)')
        prec = default_prec()
        with torch.no_grad():
            ft = self.prep(target.detach())
            if prec == lpnn.PREC_F16 and TAPS16 and lpnn.RELU_TAPE is None and all(m.weight.shape[0] % 8 == 0 for m in self.model if isinstance(m, nn.Conv2d)):
                return self._features16(ft, self._packs(prec), prec)
            return self._features(ft, self._packs(prec), prec, [])

    def forward(self, input, target, taps_t=None):
        """``taps_t``: ``target_features(target)`` computed earlier (else computed here)"""
        if not input.is_cuda:
            raise RuntimeError('PerceptualLoss runs on the MI355X HIP path only (no CPU fallback)')
        prec = default_prec()
        packs = self._packs(prec)
        fi = self.prep(input)
        if taps_t is None:
            taps_t = self.target_features(target)
        terms = self._features(fi, packs, prec, [], targets=taps_t)
        loss = torch.stack(terms).sum() if len(terms) > 1 else terms[0]     # (one cat + one sum instead of a chain of scalar adds)
        return loss * self.weight
