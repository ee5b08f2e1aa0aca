"""ResNeXt-50 32x4d and MobileNetV2 with torchvision-0.6-compatible ``state_dict`` keys, evaluated on the hand-written gfx950 kernels.

The reference instantiates both from torchvision (embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:26-28),
which is an un-vendored dependency and absent from this image; the module trees below restate the public architectures
(He et al. / Xie et al. ResNeXt; Sandler et al. MobileNetV2) as PARAMETER CONTAINERS so that reference checkpoints load by key.
They hold no arithmetic of their own:
  * ``ResNeXt.forward``      -> embedders/resnext_hip.py (``ResNeXtFunction``: forward + backward on the kernels of csrc/resnext.hip,
                                conv_dma.hip, conv_wgrad.hip), with or without autograd, train- or eval-mode BatchNorm;
  * ``MobileNetV2.forward``  -> with autograd: embedders/mobilenet_hip.py (meta-training trains the pose encoder); without: the fused fp32
                                forward ``_forward_hip`` below (fine-tuning step: the embedder is frozen and called under ``no_grad``,
                                runners/holycow.py:178-182 of the reference; drive.py);
  * a geometry the kernels do not cover, or a CPU tensor, RAISES -- there is one backend.  The stock-layer evaluation of these containers
    (the oracle of SURVEY rows E1 / E2, and what tests use for toy geometries) lives in ``oracle/backbones_ref.py``, outside the product."""
import os

import torch
import torch.nn.functional as F
from torch import nn

E_HEAD_F16_DEFAULT = ''     # layer kinds of the bf16x3 head that run with fp16 operands (ResNeXt.layer_precs)
E_F16_TAIL_DEFAULT = 6      # trailing ResNeXt bottlenecks that run with fp16 operands under the default modes (ResNeXt.block_precs)


def _unsupported(name, rule, x):
    return RuntimeError(f'{name}: input {tuple(x.shape)} on {x.device} is outside the HIP path ({rule}); there is no other backend '
                        '(tests evaluate such geometries through oracle/backbones_ref.py)')


def _padded_batch(n, ok):
    """smallest batch >= n the kernels cover (None: the image size itself is outside the path)"""
    for m in range(n, n + 9):
        if ok(m):
            return m
    return None


def _pad_rows(fn, x, n_pad):
    """fn on x with zero frames appended up to n_pad rows; -> the first len(x) rows (autograd flows through cat / narrow)"""
    n = x.shape[0]
    pad = x.new_zeros((n_pad - n,) + tuple(x.shape[1:]))
    return fn(torch.cat([x, pad]))[:n]


class _BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d as a container (same parameters, buffers and state_dict keys): the statistics, the normalisation and the running-
    statistics update happen inside the HIP encoders; ``num_batches_tracked`` of all layers advances by ONE multi-tensor add per forward."""

    def forward(self, x):
        raise RuntimeError('backbones._BatchNorm2d holds parameters and buffers only: the encoders evaluate it inside their HIP functions')


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
        raise RuntimeError('backbones._Bottleneck is a parameter container: ResNeXt.forward evaluates the whole network (embedders/resnext_hip.py)')


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
        from . import resnext_hip
        if not (x.is_cuda and x.dim() == 4 and x.shape[1] == 3):
            raise _unsupported('ResNeXt', 'N x 3 x H x W on the GPU', x)
        n, h, w = x.shape[0], x.shape[2], x.shape[3]
        if resnext_hip.supported(n, h, w):
            return self._forward_hip(x)
        # ANY batch in eval mode: the reference's fine-tuning bootstrap (train.py:241-256: training_module.eval(), no_grad,
        # b*k frames per dataloader batch) and few-shot runs hand over 1 .. 7 frames or a count that is no multiple of 4.  With running
        # statistics every frame is normalised on its own, so the batch is zero-padded to the next count the kernels cover and the extra
        # rows are dropped: EXACT (the pad frames touch no statistic and no other frame's output).  Train-mode BatchNorm couples the frames
        # of a batch -- padding would change the statistics -- so that case still raises.
        n_pad = _padded_batch(n, lambda m: resnext_hip.supported(m, h, w))
        if n_pad is None or self.training:
            raise _unsupported('ResNeXt', 'resnext_hip.supported: 32 | H, W >= 128; train-mode BatchNorm also needs N >= 8, 4 | N '
                               '(eval mode pads the batch)', x)
        return _pad_rows(self._forward_hip, x, n_pad)

    # ---- HIP path (forward and backward): embedders/resnext_hip.py -----------------------------------------------------------------
    @property
    def prec(self):
        """MFMA operand mode of the encoder's contractions (stem, classifier and every block that is not in the fp16 tail):
        LP_PREC_E (f16 | bf16x3 | bf16) if set; else bf16x3 -- ALSO when the global mode LP_PREC is f16: with train-mode BatchNorm a
        50-layer ReLU network amplifies operand rounding ~100x, and fp16 operands (2^-12) put `embeds` 7e-3 from fp64 at the 64-frame
        256 x 256 workload, bf16x3 (2^-17) 1.1e-4 (scripts/e1_parity_full.py, profiles/r04_e1_parity.txt; north_star's tolerance is 1e-3)."""
        from latent_pose_reenactment_amd.nn import PREC_NAMES, default_prec
        name = os.environ.get('LP_PREC_E')
        if name:
            return PREC_NAMES[name]
        return PREC_NAMES['bf16x3'] if default_prec() == PREC_NAMES['f16'] else default_prec()

    def block_precs(self):
        """per-bottleneck operand mode: the LAST ``LP_E_F16_TAIL`` blocks run with fp16 operands (1 MFMA per MAC, 16-bit-resident conv
        outputs) when the encoder's mode is bf16x3 under the global fp16 mode -- rounding injected late meets fewer BatchNorm layers
        (the sweep behind the default: profiles/r04_e1_parity.txt)."""
        from latent_pose_reenactment_amd.nn import PREC_NAMES, default_prec
        self._hip_structure()
        nb = len(self._hip_blocks)
        base = self.prec
        tail = 0
        if base == PREC_NAMES['bf16x3'] and default_prec() == PREC_NAMES['f16'] and not os.environ.get('LP_PREC_E'):
            tail = max(0, min(nb, int(os.environ.get('LP_E_F16_TAIL', str(E_F16_TAIL_DEFAULT)))))
        return [base] * (nb - tail) + [PREC_NAMES['f16']] * tail

    def layer_precs(self):
        """operand mode per contraction: {(block name, 'conv1' | 'conv2' | 'conv3'): mode}; the downsample conv shares conv1's input planes
        and mode.  ``LP_E_HEAD_F16`` = comma list of layer kinds that run with fp16 operands ALSO in the bf16x3 blocks of the default
        assignment (an experiment knob, empty by default: profiles/r04_e1_parity.txt has the sweep)."""
        from latent_pose_reenactment_amd.nn import PREC_NAMES, default_prec
        self._hip_structure()
        kinds = set(k for k in os.environ.get('LP_E_HEAD_F16', E_HEAD_F16_DEFAULT).split(',') if k)
        assert kinds <= {'conv1', 'conv2', 'conv3'}, kinds
        mixed = bool(kinds) and default_prec() == PREC_NAMES['f16'] and not os.environ.get('LP_PREC_E')
        out = {}
        for (bname, *_), p in zip(self._hip_blocks, self.block_precs()):
            for kind in ('conv1', 'conv2', 'conv3'):
                out[(bname, kind)] = PREC_NAMES['f16'] if (mixed and p == PREC_NAMES['bf16x3'] and kind in kinds) else p
        return out

    def _hip_structure(self):
        if self.__dict__.get('_hip_param_names') is None:
            self.__dict__['_hip_param_names'] = [k for k, _ in self.named_parameters()]
            self.__dict__['_hip_bn'] = {k: m for k, m in self.named_modules() if isinstance(m, nn.BatchNorm2d)}
            blocks = []
            for li in range(1, 5):
                for bi, blk in enumerate(getattr(self, f'layer{li}')):
                    blocks.append((f'layer{li}.{bi}', blk.conv1.in_channels, blk.conv2.in_channels, blk.conv3.out_channels,
                                   blk.conv2.stride[0], blk.downsample is not None))
            self.__dict__['_hip_blocks'] = blocks

    def _hip_packs(self, par, need_grad):
        """16-bit weight packs: name -> (forward pack, data-gradient pack | None).  The dense contractions (1x1 convs, the stem viewed as
        [64][147], the classifier) are re-packed by one batched launch per orientation set (static buffers: hipGraph friendly); the
        grouped 3x3 weights by lp_pack_grouped.  Without autograd and in eval mode the packs are cached until a weight changes."""
        from latent_pose_reenactment_amd import hipops as ops
        from latent_pose_reenactment_amd.optim import WEIGHTS_GENERATION
        base = self.prec
        bprec = self.layer_precs()

        def prec_of(k):          # 'layer3.4.conv1.weight' -> the mode of that contraction ('...downsample.0.weight': conv1's); stem and classifier: the base mode
            part = k.split('.')
            return bprec.get(('.'.join(part[:2]), 'conv1' if part[2] == 'downsample' else part[2]), base) if len(part) > 3 else base
        dense = [k for k in self._hip_param_names
                 if k == 'conv1.weight' or k == 'fc.weight' or (par[k].dim() == 4 and par[k].shape[2] == 1)]      # stem, classifier, 1x1 convs
        grouped = [k for k in self._hip_param_names if par[k].dim() == 4 and par[k].shape[2] == 3]                 # the 16 grouped 3x3 convs
        cacheable = not need_grad and not self.training
        key = (tuple(sorted(bprec.items())), base, need_grad, WEIGHTS_GENERATION[0]) + tuple((par[k].data_ptr(), par[k]._version) for k in dense + grouped)
        cache = self.__dict__.get('_hip_pack_cache')
        if cacheable and cache is not None and cache[0] == key:
            return cache[1]

        def w2d(k):
            w = par[k].detach()
            return w.view(w.shape[0], -1) if k == 'conv1.weight' else w
        packs = {}
        pbs = self.__dict__.setdefault('_hip_pbs', {})
        for prec in sorted(set(prec_of(k) for k in dense)):          # one batched re-pack per precision mode in use
            dk = [k for k in dense if prec_of(k) == prec]
            specs = [(w2d(k), 0, False) for k in dk]
            if need_grad:
                specs += [(w2d(k), 1, False) for k in dk if k != 'conv1.weight']
            pb = pbs.get(prec)
            pkey = tuple((w.data_ptr(), m, bool(sk)) for w, m, sk in specs)
            if pb is None or pb.prec != prec or pb.key != pkey:
                pb = ops.PackBatch(specs, prec)
                pbs[prec] = pb
            allp = pb.update()
            for i, k in enumerate(dk):
                packs[k] = [allp[i], None]
            if need_grad:
                for j, k in enumerate(k_ for k_ in dk if k_ != 'conv1.weight'):
                    packs[k][1] = allp[len(dk) + j]
        for k in grouped:
            w = par[k].detach().contiguous()
            packs[k] = [ops.pack_grouped(w, 0, prec_of(k)), ops.pack_grouped(w, 1, prec_of(k)) if need_grad else None]
        if cacheable:
            self.__dict__['_hip_pack_cache'] = (key, packs)
        return packs

    def _hip_eval_affines(self, par):
        """eval mode: every BatchNorm as (mean, rstd, scale, shift) from its running statistics (five multi-tensor ops for all layers)"""
        from .resnext_hip import _BN
        names = list(self._hip_bn)
        bns = [self._hip_bn[k] for k in names]
        rstd = torch._foreach_add([m.running_var for m in bns], bns[0].eps)
        torch._foreach_rsqrt_(rstd)
        sc = torch._foreach_mul(rstd, [par[k + '.weight'].detach() for k in names])
        sh = torch._foreach_mul([m.running_mean for m in bns], sc)
        sh = torch._foreach_sub([par[k + '.bias'].detach() for k in names], sh)
        return {k: _BN(m.running_mean, r, a_, b_) for k, m, r, a_, b_ in zip(names, bns, rstd, sc, sh)}

    def _forward_hip(self, x):
        from .resnext_hip import ResNeXtFunction
        self._hip_structure()
        return ResNeXtFunction.apply(self, x, *[p for _, p in self.named_parameters()])


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
        raise RuntimeError('backbones._InvertedResidual is a parameter container: MobileNetV2.forward evaluates the whole network')


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
        if not (x.is_cuda and x.dim() == 4 and x.shape[1] == 3):
            raise _unsupported('MobileNetV2', 'N x 3 x H x W on the GPU', x)
        train_bn = self.features[0][1].training
        if not torch.is_grad_enabled():
            if x.shape[2] % 2 or x.shape[3] % 2:
                raise _unsupported('MobileNetV2 (no-grad forward)', 'even H, W', x)
            if x.shape[0] > 64:
                #  the fused forward takes <= 64 frames per launch sequence: larger eval batches run as chunks (exact: running
                # statistics); train-mode batch statistics would differ per chunk, so that case raises
                if train_bn:
                    raise _unsupported('MobileNetV2 (no-grad forward, train-mode BatchNorm)', 'N <= 64', x)
                return torch.cat([self._forward_hip(x[i:i + 64]) for i in range(0, x.shape[0], 64)])
            return self._forward_hip(x)                      # fused fp32 forward (fine-tuning step, drive.py)
        from . import mobilenet_hip
        n, h, w = x.shape[0], x.shape[2], x.shape[3]
        if not mobilenet_hip.supported(n, h, w):
            n_pad = _padded_batch(n, lambda m: mobilenet_hip.supported(m, h, w))
            if n_pad is None or train_bn:
                raise _unsupported('MobileNetV2 (autograd on)', 'mobilenet_hip.supported: 32 | H, W; train-mode BatchNorm also needs '
                                   'N * H/32 * W/32 >= 8 and a multiple of 4 (eval mode pads the batch)', x)
            return _pad_rows(self._forward_hip_train, x, n_pad)          # (eval-mode BatchNorm: zero frames appended, their rows dropped -- exact)
        return self._forward_hip_train(x)                    # autograd on (meta-training): forward + backward on the HIP kernels

    # ---- HIP training path (forward + backward): embedders/mobilenet_hip.py --------------------------------------------------------
    def _hip_train_structure(self):
        if self.__dict__.get('_hip_feature_param_names') is None:
            self.__dict__['_hip_feature_param_names'] = [k for k, _ in self.features.named_parameters()]
            self.__dict__['_hip_feature_bn'] = {k: m for k, m in self.features.named_modules() if isinstance(m, nn.BatchNorm2d)}

    def _hip_train_packs(self, par, need_grad):
        """bf16x3 packs of the dense contractions (stem viewed as [32][27], every 1x1 conv): name -> [forward, data-gradient | None];
        one batched launch per call in training, cached for no-grad eval calls"""
        from latent_pose_reenactment_amd import hipops as ops
        from latent_pose_reenactment_amd.optim import WEIGHTS_GENERATION
        from .mobilenet_hip import PREC
        dense = [k for k in self._hip_feature_param_names if k == '0.0.weight' or (par[k].dim() == 4 and par[k].shape[2] == 1)]
        cacheable = not need_grad and not self.training
        key = (need_grad, WEIGHTS_GENERATION[0]) + tuple((par[k].data_ptr(), par[k]._version) for k in dense)
        cache = self.__dict__.get('_hip_train_pack_cache')
        if cacheable and cache is not None and cache[0] == key:
            return cache[1]

        def w2d(k):
            w = par[k].detach()
            return w.view(w.shape[0], -1) if k == '0.0.weight' else w
        specs = [(w2d(k), 0, False) for k in dense]
        if need_grad:
            specs += [(w2d(k), 1, False) for k in dense if k != '0.0.weight']
        pb = self.__dict__.get('_hip_train_pb')
        pkey = tuple((w.data_ptr(), m, bool(sk)) for w, m, sk in specs)
        if pb is None or pb.key != pkey:
            pb = ops.PackBatch(specs, PREC)
            self.__dict__['_hip_train_pb'] = pb
        allp = pb.update()
        packs = {k: [allp[i], None] for i, k in enumerate(dense)}
        if need_grad:
            for j, k in enumerate(k_ for k_ in dense if k_ != '0.0.weight'):
                packs[k][1] = allp[len(dense) + j]
        if cacheable:
            self.__dict__['_hip_train_pack_cache'] = (key, packs)
        return packs

    def _hip_eval_bn(self, par):
        from .resnext_hip import _BN
        names = list(self._hip_feature_bn)
        bns = [self._hip_feature_bn[k] for k in names]
        rstd = torch._foreach_add([m.running_var for m in bns], bns[0].eps)
        torch._foreach_rsqrt_(rstd)
        sc = torch._foreach_mul(rstd, [par[k + '.weight'].detach() for k in names])
        sh = torch._foreach_mul([m.running_mean for m in bns], sc)
        sh = torch._foreach_sub([par[k + '.bias'].detach() for k in names], sh)
        return {k: _BN(m.running_mean, r, a_, b_) for k, m, r, a_, b_ in zip(names, bns, rstd, sc, sh)}

    def _forward_hip_train(self, x):
        from .mobilenet_hip import LinearRowsFunction, MobileNetFeaturesFunction
        self._hip_train_structure()
        pooled = MobileNetFeaturesFunction.apply(self, x, *[p for _, p in self.features.named_parameters()])
        drop, fc = self.classifier[0], self.classifier[1]
        pooled = F.dropout(pooled, drop.p, drop.training)
        return LinearRowsFunction.apply(pooled, fc.weight, fc.bias)

    # ---- HIP forward (no autograd) ---------------------------------------------------------------------------------------------
    def _eval_affines(self):
        """eval mode: every BatchNorm folded to a per-channel (scale, shift) by four multi-tensor ops, cached until a weight or
        buffer changes (``_version``, or the generation counter of the fused optimizer / EMA kernels)"""
        from latent_pose_reenactment_amd.optim import WEIGHTS_GENERATION
        # (the tensor OBJECTS are re-collected on every call: Module._apply -- .to / .cuda / .float -- replaces buffers, and a key built
        #  from stale objects would miss later in-place updates of the new ones)
        tens = [t for m in self.modules() if isinstance(m, nn.BatchNorm2d) for t in (m.weight, m.bias, m.running_mean, m.running_var)]
        key = (WEIGHTS_GENERATION[0], sum(t._version for t in tens), tuple(t.data_ptr() for t in tens[:8]))
        st = self.__dict__.get('_hip_cache')
        if st is not None and st[0] == key:
            return st[1]
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        sc = torch._foreach_add([m.running_var for m in bns], bns[0].eps)
        torch._foreach_rsqrt_(sc)
        torch._foreach_mul_(sc, [m.weight.detach() for m in bns])
        sh = torch._foreach_mul([m.running_mean for m in bns], sc)
        sh = torch._foreach_sub([m.bias.detach() for m in bns], sh)
        affines = {m: (a_, b_) for m, a_, b_ in zip(bns, sc, sh)}
        self.__dict__['_hip_cache'] = (key, affines)
        return affines

    def _forward_hip(self, x):
        """fp32 throughout.  A conv launch leaves its RAW output; the BatchNorm that follows is a per-channel (scale, shift) which
        the NEXT launch applies while it loads that tensor (with ReLU6, or with the block's residual add).  Train mode: the conv
        launch also leaves the statistics partials of its output and lp_bn_finalize turns them into (scale, shift) and the
        running-statistics update -- two launches per conv + BatchNorm (+ ReLU6 / residual); eval mode: one."""
        from latent_pose_reenactment_amd import hipops as ops
        feats = list(self.features)
        train = feats[0][1].training
        affines = None if train else self._eval_affines()
        counters = []

        def bn(m, stats):
            if not train:
                return affines[m]
            if m.track_running_stats:
                counters.append(m.num_batches_tracked)
            return ops.bn_finalize(stats, m.weight.detach(), m.bias.detach(), m.running_mean, m.running_var, m.momentum, m.eps)

        x = x.contiguous().float()
        y = ops.stem_conv_s2(x, feats[0][0].weight.detach())
        if train:
            if feats[0][1].track_running_stats:
                counters.append(feats[0][1].num_batches_tracked)
            m0 = feats[0][1]
            aff = ops.bn_batch_affine(y, m0.weight.detach(), m0.bias.detach(), m0.running_mean, m0.running_var, m0.momentum, m0.eps)
        else:
            aff = affines[feats[0][1]]
        pend = (y, aff)                                  # raw conv output + the BatchNorm-ReLU6 its consumer applies while loading
        cur = None                                       # block input: (raw project output, its BatchNorm (scale, shift), residual | None)
        for blk in feats[1:-1]:
            layers = list(blk.conv)
            res = None
            if len(layers) == 4:
                raw, (s, t), r = cur
                ye, xin, st = ops.pwconv(raw, layers[0][0].weight.detach(), in_scale=s, in_shift=t, in_res=r, want_x=blk.use_res, stats=train)
                pend = (ye, bn(layers[0][1], st))
                res = xin                                # x = BN(raw) + r, written back by the expand launch when the block adds it later
            dw, pw, pbn = layers[-3], layers[-2], layers[-1]
            if train:
                yd, st = ops.dwconv3x3_stats(pend[0], dw[0].weight.detach(), dw[0].stride[0], pend[1][0], pend[1][1])
            else:
                yd, st = ops.dwconv3x3(pend[0], dw[0].weight.detach(), dw[0].stride[0], pend[1][0], pend[1][1]), None
            s, t = bn(dw[1], st)
            yp, _, st = ops.pwconv(yd, pw.weight.detach(), in_scale=s, in_shift=t, in_relu6=True, stats=train)
            cur = (yp, bn(pbn, st), res if blk.use_res else None)
        last = feats[-1]
        raw, (s, t), r = cur
        yl, _, st = ops.pwconv(raw, last[0].weight.detach(), in_scale=s, in_shift=t, in_res=r, stats=train)
        s, t = bn(last[1], st)
        pooled = ops.affine_relu6_mean(yl, s, t)
        if counters:
            torch._foreach_add_(counters, 1)
        drop, fc = self.classifier[0], self.classifier[1]
        pooled = F.dropout(pooled, drop.p, drop.training)
        return ops.linear_fwd(pooled, fc.weight.detach(), fc.bias.detach(), None)


def mobilenet_v2(num_classes=1000):
    return MobileNetV2(num_classes)
