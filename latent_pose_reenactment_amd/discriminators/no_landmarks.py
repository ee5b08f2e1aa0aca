"""Projection-discriminator plugin (reference API: discriminators/no_landmarks.py:11-166), key-compatible state_dict.

Behavioural notes reproduced on purpose:
  * every norm-'none' ResBlock starts with ReLU(inplace) on its *input* (generators/common/blocks.py:71-73), so its skip
    branch / identity add see relu(x), and the features handed to feature matching are post-ReLU except the last one;
  * pass_inputs runs three times per step (fake->G, fake.detach->D, real), each doing its own power iteration.
All convolutions run on the gfx950 kernels (lp_act_pack / lp_conv16_fwd with bias / residual epilogue, lp_conv16_wgrad,
lp_avgpool2_*); activations are NHWC inside and are handed out as logical N x C x H x W views (channels_last strides).
avgpool(a) + avgpool(b) is evaluated as avgpool(a + b) (the skip conv result is the residual operand of conv2's epilogue):
identical algebra, one rounding fewer."""
import math

import torch
import torch.nn.functional as F
from torch import nn

from latent_pose_reenactment_amd import hipops as ops
from latent_pose_reenactment_amd import nn as lpnn
from latent_pose_reenactment_amd.nn import (SNWeight, SNBatch, SNLinearFn, SNEmbeddingFn, _Indexed, SN_EPS_CONV, SN_EPS_DEFAULT, AvgPool2Fn,
                                            as_nchw_view, default_prec, hip_conv, to_nhwc)
from latent_pose_reenactment_amd.utils import radam as _radam

torch.optim.RAdam = _radam.RAdam      # the reference swaps in its own RAdam the same way (no_landmarks.py:5-6)
import os as _os
# conv -> ReLU -> conv chains: the first conv writes only the operand planes of relu(h) (round 4; 0: also the fp32 h nobody reads)
PLANES_ONLY = _os.environ.get('LP_D_PLANES_ONLY', '1') != '0'
# pool -> (next block's in-place) ReLU -> operand planes in ONE launch (round 5; 0: three launches)
FUSE_POOL_RELU = _os.environ.get('LP_D_POOL_RELU', '1') != '0'
# conv2 + skip + AvgPool2d(2) + the next block's ReLU of every down block / the stem as ONE 4x4 stride-2 conv launch (round 6, nn.ConvPoolFn: 4/9 of the
# matrix work, no full-resolution block output); 0: conv, then the pool launch
CONV_POOL = _os.environ.get('LP_D_CONVPOOL', '1') != '0'


def gpass_prec():
    """operand mode of the fake -> G pass (the pass whose score IS a loss scalar: adversarial_G = -mean(fake_score_G),
    criterions/adversarial.py:34-57 of the reference).  The projection score <pooled, embed> + linear(pooled) is the dot product of a
    512-vector with an unrelated direction, ~sqrt(512) smaller than the norms it is made of, so the feature error of an fp16 critic
    (3e-4) shows up ~15x larger relative to the score (5e-3 measured at the full configs[2] geometry, round 4) -- outside north_star's
    1e-3 on every loss scalar.  Under the global fp16 mode this pass therefore runs its blocks from ``LP_D_GPASS_FROM`` on (default 0: all
    of it; its weights are constants for autograd, so that is forward + data gradient only) with bf16x3 operands.  LP_D_GPASS_PREC=f16
    restores round 4's all-fp16 pass.  -> (mode, first strict unit: 0 = stem, 1.. = blocks)"""
    name = _os.environ.get('LP_D_GPASS_PREC', 'bf16x3')
    if default_prec() != lpnn.PREC_F16 or name == 'f16':
        return default_prec(), 0
    return lpnn.PREC_NAMES[name], int(_os.environ.get('LP_D_GPASS_FROM', '0'))


def dpass_prec():
    """operand mode of the two discriminator-side passes (fake.detach -> D, real -> D: the passes behind loss_D.backward, i.e. behind every
    parameter gradient the critic's optimizer sees).  The hinge loss puts -1/2 on the real and +1/2 on the fake sample, so the weight gradients
    of the last blocks are DIFFERENCES of nearly equal terms and fp16 operand rounding shows up amplified in them (9.3e-3 tie-masked on
    ``blocks.5.block.5`` at 256 x 256, round 5) -- outside SURVEY 8d's 1e-3 on every parameter gradient.  ``LP_D_DPASS_PREC`` (f16 | bf16x3)
    and ``LP_D_DPASS_FROM`` (first strict unit: 0 = stem, 1.. = blocks) choose the assignment; -> (mode, first strict unit)"""
    name = _os.environ.get('LP_D_DPASS_PREC', DPASS_DEFAULT[0])
    if default_prec() != lpnn.PREC_F16 or name == 'f16':
        return default_prec(), 0
    return lpnn.PREC_NAMES[name], int(_os.environ.get('LP_D_DPASS_FROM', str(DPASS_DEFAULT[1])))


DPASS_DEFAULT = ('bf16x3', 3)


class Wrapper:
    @staticmethod
    def get_args(parser):
        parser.add('--dis_padding', type=str, default='zero', help='zero|reflection')
        parser.add('--dis_num_blocks', type=int, default=7)
        parser.add('--lr_dis', type=float, default=2e-4)

    @staticmethod
    def get_net(args):
        return Discriminator(args.dis_padding, args.in_channels, args.out_channels, args.num_channels, args.max_num_channels,
                             args.embed_channels, args.dis_num_blocks, args.image_size, args.num_labels).to(args.device)

    @staticmethod
    def get_optimizer(discriminator, args):
        from runners.holycow import optimizer_class
        return optimizer_class(args.optimizer, args.device)(discriminator.parameters(), lr=args.lr_dis, betas=(args.beta1, 0.999), eps=1e-5)


def _wb(layer, track, states):
    """(W_orig, bias, sn-state) of an SN conv for ``hip_conv``; ``track=False`` detaches the parameters from autograd"""
    w, b = layer.weight_orig, layer.bias
    if not track:
        w, b = w.detach(), (None if b is None else b.detach())
    return w, b, states[id(layer)]


def _conv_pool(h, layer, track, states, prec, res, x16, relu_out, emit):
    """conv3x3(relu(h)) + bias, AvgPool2d(2), + res, [ReLU] in one launch (nn.ConvPoolFn); packs shared through the per-step cache like ``_conv``"""
    w, b, st = _wb(layer, track, states)
    prec = default_prec() if prec is None else prec
    return lpnn.ConvPoolFn.apply(h, w, b, res, prec, states['packs'].setdefault(prec, {}), st, x16, relu_out, emit)


def _conv(x, layer, track, states, prec=None, **kw):
    """SN conv through the HIP kernels; the 16-bit packs of W_orig are shared by the three passes of a step and their backward
    passes (only 1/sigma differs between passes) through the per-step cache ``states['packs']`` (one dict per operand mode)"""
    w, b, st = _wb(layer, track, states)
    prec = default_prec() if prec is None else prec
    return hip_conv(x, w, b, sn=st, packs=states['packs'].setdefault(prec, {}), prec=prec, **kw)


class _DisBlock(nn.Module):
    """parameters of blocks.ResBlock(norm_layer='none'): block.2, block.5 (3x3 + bias) and optional skip.0 (1x1 + bias)"""

    def __init__(self, cin, cout, downsample, reflect=False):
        super().__init__()
        self.reflect = reflect          # dis_padding='reflection': nn.ReflectionPad2d(1) in front of the block's two 3x3 convs (blocks.py:76-88)
        self.block = _Indexed(_2=SNWeight((cout, cin, 3, 3), True, SN_EPS_CONV), _5=SNWeight((cout, cout, 3, 3), True, SN_EPS_CONV))
        self.has_skip = cin != cout or downsample
        if self.has_skip:
            self.skip = _Indexed(_0=SNWeight((cout, cin, 1, 1), True, SN_EPS_CONV))
        self.downsample = downsample

    def sn_layers(self):
        return [self.block._modules['2'], self.block._modules['5']] + ([self.skip._modules['0']] if self.has_skip else [])

    def forward(self, x_relu, track, states, prec=None, xr16=None, next_prec=None, last=False):
        """x_relu: NHWC relu(x) (the reference's in-place ReLU makes every consumer of the block input see relu(x)); ``xr16``: its operand
        planes when the producer already wrote them.  A down-sampling block returns ``(relu(pool(out)), planes | None)``: the in-place ReLU
        the NEXT block applies (blocks.py:71-73) and the next conv's operand planes (mode ``next_prec``) come from the pool launch itself
        (FUSE_POOL_RELU); the last block returns its pre-activation output."""
        c1, c2 = self.block._modules['2'], self.block._modules['5']
        prec = default_prec() if prec is None else prec
        # relu(x) is packed to operand planes ONCE for its two consumers; conv1's epilogue emits the planes of relu(h) for conv2
        if xr16 is None:
            xr16 = ops.act_pack(x_relu, pro=0, prec=prec)
        if self.reflect:          # (border terms are added after each conv launch: no epilogue-emitted planes; conv2 packs relu(h) itself)
            h = _conv(x_relu, c1, track, states, prec, ksize=3, x16=xr16, reflect=True)
            shortcut = _conv(x_relu, self.skip._modules['0'], track, states, prec, ksize=1, x16=xr16) if self.has_skip else x_relu
            out = _conv(h, c2, track, states, prec, res=shortcut, ksize=3, pro=2, reflect=True)
            if self.downsample and last:
                return AvgPool2Fn.apply(out, False)
            return pool_relu(out, next_prec) if self.downsample else out
        h, h16 = _conv(x_relu, c1, track, states, prec, ksize=3, x16=xr16, emit16=1, want_y=not PLANES_ONLY)      # (h itself is never read: conv2 takes the planes)
        if self.downsample and CONV_POOL:
            # pool(conv2(relu(h)) + skip(x)) = conv4x4/2(relu(h)) + skip(pool(x)): the 1x1 skip conv commutes with the average
            xp = AvgPool2Fn.apply(x_relu, False)
            shortcut_lo = _conv(xp, self.skip._modules['0'], track, states, prec, ksize=1) if self.has_skip else xp
            nprec = default_prec() if next_prec is None else next_prec
            holder = [] if (not last and nprec == prec) else None          # (the epilogue emits planes in its own operand mode only)
            y = _conv_pool(h, c2, track, states, prec, shortcut_lo, h16, not last, holder)
            return y if last else (y, holder[0] if holder else None)
        shortcut = _conv(x_relu, self.skip._modules['0'], track, states, prec, ksize=1, x16=xr16) if self.has_skip else x_relu
        out = _conv(h, c2, track, states, prec, res=shortcut, ksize=3, pro=2, x16=h16)
        if self.downsample and last:          # the final feature map is handed out BEFORE any ReLU (the reference appends it un-mutated)
            return AvgPool2Fn.apply(out, False)
        return pool_relu(out, next_prec) if self.downsample else out


def pool_relu(out, next_prec):
    """-> (relu(AvgPool2d(2)(out)), its operand planes in mode ``next_prec`` | None) in one launch; LP_D_POOL_RELU=0: pool, then torch.relu"""
    next_prec = default_prec() if next_prec is None else next_prec
    if not FUSE_POOL_RELU:
        return torch.relu(AvgPool2Fn.apply(out, False)), None
    holder = []
    y = AvgPool2Fn.apply(out, False, (next_prec, holder), True)
    o16 = holder[0] if holder else None
    # (the pool launch writes one-plane operand modes; for a bf16x3 consumer hipops.avgpool2_fwd falls back to an act_pack of y)
    return y, o16


class Discriminator(nn.Module):
    def __init__(self, padding, in_channels, out_channels, num_channels, max_num_channels, embed_channels, dis_num_blocks,
                 image_size, num_labels):
        super().__init__()
        if padding not in ('zero', 'reflection'):
            raise Exception('Incorrect `padding` argument, required `zero` or `reflection`')       # (no_landmarks.py:49-50)
        reflect = padding == 'reflection'          # (the stem's convs keep their zero padding: no_landmarks.py:52-60 has `padding(1)` commented out)
        self.out_channels = embed_channels
        self.down_block = _Indexed(_0=SNWeight((num_channels, in_channels, 3, 3), True, SN_EPS_CONV),
                                   _2=SNWeight((num_channels, num_channels, 3, 3), True, SN_EPS_CONV))
        self.skip = _Indexed(_0=SNWeight((num_channels, in_channels, 1, 1), True, SN_EPS_CONV))
        self.blocks = nn.ModuleList()
        num_down = min(int(math.log(image_size, 2)) - 2, dis_num_blocks)
        cin = num_channels
        cout = cin
        for i in range(1, num_down):
            cout = min(cin * 2, max_num_channels)
            if i == dis_num_blocks - 1:
                cout = self.out_channels
            self.blocks.append(_DisBlock(cin, cout, True, reflect))
            cin = cout
        for i in range(num_down, dis_num_blocks):
            if i == dis_num_blocks - 1:
                cout = self.out_channels
            self.blocks.append(_DisBlock(cin, cout, False, reflect))
        self.linear = SNWeight((1, self.out_channels), True, SN_EPS_CONV)
        self.embed = SNWeight((num_labels, self.out_channels), False, SN_EPS_CONV)
        with torch.no_grad():
            self.embed.weight_orig.uniform_(-0.1, 0.1)
        self.finetuning = False

    def _fresh_packs(self):
        """16-bit packs (forward + dgrad) of every conv of the critic for this step's three passes, one batched launch per operand mode in
        use (the default mode; the fake -> G pass's strict mode for the layers it covers: ``gpass_prec``) -> {mode: {(W_orig.data_ptr(),
        orientation): pack}}, the inner dicts keyed like ConvFn's per-step cache"""
        d0, d2, sk = self.down_block._modules['0'], self.down_block._modules['2'], self.skip._modules['0']
        units = [[d0, d2, sk]] + [blk.sn_layers() for blk in self.blocks]          # unit 0 = stem, 1.. = blocks (gpass_prec's numbering)
        if not d0.weight_orig.is_cuda:
            return {}
        training = self.training and torch.is_grad_enabled()
        out = {}
        strict_from = {}          # operand mode other than the default one -> first unit any pass runs in it
        for sprec, sfrom in (gpass_prec(), dpass_prec()):
            if sprec != default_prec() and training:
                strict_from[sprec] = min(sfrom, strict_from.get(sprec, sfrom))
        for prec, convs in [(default_prec(), [m for u in units for m in u])] + [(sp, [m for u in units[sf:] for m in u]) for sp, sf in strict_from.items()]:
            if not convs:
                continue
            # conv + pool launches (nn.ConvPoolFn: the stem's second conv, conv2 of every down block) take the 16-tap images of modes 4 / 5
            pooled = {id(d2)} | {id(blk.block._modules['5']) for blk in self.blocks if blk.downsample and not blk.reflect} if CONV_POOL else set()
            specs = []
            for m in convs:
                w = m.weight_orig
                ks = w.shape[-1]
                if id(m) in pooled:
                    specs += [(w, 4, False), (w, 5, False)]
                    continue
                specs.append((w, 0, ks == 3 and w.shape[1] <= 32))
                specs.append((w, 1, ks == 3 and w.shape[0] <= 32))
            key = tuple((w.data_ptr(), mode, bool(k_)) for w, mode, k_ in specs)
            pbs = self.__dict__.setdefault('_pack_batches', {})
            pb = pbs.get(prec)
            if pb is None or pb.prec != prec or pb.key != key:
                pb = ops.PackBatch([(w.detach(), mode, k_) for w, mode, k_ in specs], prec)
                pbs[prec] = pb
            packs = pb.update()
            out[prec] = {(w.data_ptr(), mode): p for (w, mode, _), p in zip(specs, packs)}
        return out

    def _conv_sn_layers(self):
        layers = self.__dict__.get('_sn_layer_cache')
        if layers is None:
            d0, d2, sk = self.down_block._modules['0'], self.down_block._modules['2'], self.skip._modules['0']
            layers = [d0, d2, sk] + [l for blk in self.blocks for l in blk.sn_layers()] + [self.linear]
            self.__dict__['_sn_layer_cache'] = layers
            self.__dict__['_sn_batch'] = SNBatch(layers)
        return layers

    def _embed_batch(self):
        esn = self.__dict__.get('_embed_sn')
        if esn is None or esn.layers[0] is not self.embed:
            esn = SNBatch([self.embed])
            self.__dict__['_embed_sn'] = esn
        return esn

    def prepare_step(self):
        """The parts of a TRAINING forward that depend on the weights only -- the 16-bit weight packs, the power iteration of the label
        embedding and the power iterations of the three passes -- may be issued ahead of ``forward`` (on a side stream, beside the
        encoders: runners/holycow.py, streams.py).  ``forward`` / ``pass_inputs`` pick the results up in order.  Same arithmetic either way."""
        if not (self.training and torch.is_grad_enabled() and next(self.parameters()).is_cuda):
            return
        self._conv_sn_layers()
        self.__dict__['_prepared'] = (self._fresh_packs(), self._embed_batch().update(True)[0])
        self.__dict__['_prepared_passes'] = [self._sn_batch.update(True) for _ in range(3)]

    def pass_inputs(self, x, embed=None, track_weights=True, sn_states=None, strict=None):
        """``track_weights=False``: the discriminator's own parameters are constants for autograd in this pass (gradients
        still flow to ``x`` and ``embed``).  ``strict`` = (operand mode, first unit) from ``gpass_prec``: the stem (unit 0) / blocks
        (1..) from that unit on run in that operand mode instead of the default one (the fake -> G pass)."""
        if not x.is_cuda:
            raise RuntimeError('the discriminator runs on the MI355X HIP path only (no CPU fallback)')
        d0, d2, sk = self.down_block._modules['0'], self.down_block._modules['2'], self.skip._modules['0']
        # one launch power-iterates all spectrally normalised layers of this pass (every pass does its own iteration; prepare_step() may
        # have run the three iterations of this step ahead)
        layers = self._conv_sn_layers()
        ahead = self.__dict__.get('_prepared_passes')
        st = sn_states if sn_states is not None else (ahead.pop(0) if ahead else self._sn_batch.update(self.training))
        states = {id(l): s for l, s in zip(layers, st)}
        states['packs'] = self.__dict__.setdefault('_step_packs', {})
        xn = to_nhwc(x)
        sprec, sfrom = strict if strict is not None else (None, 0)
        unit_prec = lambda u: sprec if (sprec is not None and u >= sfrom) else None          # None = the default mode
        p0 = unit_prec(0)
        h, h16 = _conv(xn, d0, track_weights, states, p0, ksize=3, emit16=1, want_y=not PLANES_ONLY)
        nb = len(self.blocks)
        if CONV_POOL:
            # stem: pool(conv2(relu(h)) + skip(image)) as one launch; the skip conv (3 -> 64, 1x1) runs on the pooled image
            shortcut_lo = _conv(to_nhwc(F.avg_pool2d(x, 2)), sk, track_weights, states, p0, ksize=1)
            pp0 = default_prec() if p0 is None else p0
            np1 = unit_prec(1) if nb else None
            holder = [] if (default_prec() if np1 is None else np1) == pp0 else None
            out_relu = _conv_pool(h, d2, track_weights, states, p0, shortcut_lo, h16, True, holder)
            xr16 = holder[0] if holder else None
        else:
            shortcut = _conv(xn, sk, track_weights, states, p0, ksize=1)
            out_relu, xr16 = pool_relu(_conv(h, d2, track_weights, states, p0, res=shortcut, ksize=3, pro=2, x16=h16), unit_prec(1) if nb else None)
        feats = []
        out = out_relu
        for bi, block in enumerate(self.blocks):
            # out_relu = relu(previous unit's output): what the reference's in-place ReLU leaves behind in its feature list
            lpnn.tape_relu(lambda: out_relu > 0)
            feats.append(as_nchw_view(out_relu))
            last = bi + 1 == nb
            res = block(out_relu, track_weights, states, unit_prec(bi + 1), xr16=xr16, next_prec=None if last else unit_prec(bi + 2), last=last)
            if block.downsample and not last:
                out_relu, xr16 = res
                out = out_relu
            else:
                out = res
                if bi + 1 < nb:          # (a non-pooling block in the middle of the stack: its successor's in-place ReLU)
                    out_relu, xr16 = torch.relu(out), None
        feats.append(as_nchw_view(out))
        lpnn.tape_relu(lambda: out > 0)
        pooled, dot = lpnn.ProjScoreFn.apply(out, embed)          # sum_hw relu(out) and <pooled, embed> in one launch
        wl, bl, sl = _wb(self.linear, track_weights, states)
        score = SNLinearFn.apply(pooled, wl, bl, *sl)[:, 0]
        if embed is not None:
            score = dot + score
        return score, feats

    def enable_finetuning(self, data_dict=None):
        """no_landmarks.py:110-136: the label embedding matrix collapses to one row holding the identity embedding."""
        ref = next(iter(self.parameters()))
        if data_dict is None:
            data_dict = {'embeds': torch.rand(1, self.out_channels).to(ref)}
        with torch.no_grad():
            if self.finetuning:
                self.embed.weight_orig.copy_(data_dict['embeds'])
            else:
                fresh = SNWeight((1, self.out_channels), False, SN_EPS_DEFAULT).to(ref)
                fresh.weight_orig.copy_(data_dict['embeds'])
                self.embed = fresh
                self.finetuning = True

    def start_real_pass(self, data_dict):
        """streams 'real' (LP_OVERLAP_REAL, on in meta-training; runners/holycow.py calls this between the encoders and the generator): the pass over the REAL image depends on
        neither encoder nor generator -- with the weights prepared ahead (``prepare_step``) it can be issued on its side stream beside the
        generator's forward instead of beside the other two passes.  Same arithmetic: it still uses the third power iteration of the step."""
        from latent_pose_reenactment_amd import streams
        real, label = data_dict.get('target_rgbs'), data_dict.get('label')
        prepared, ahead = self.__dict__.get('_prepared'), self.__dict__.get('_prepared_passes')
        if (real is None or label is None or prepared is None or not ahead or len(ahead) != 3 or lpnn.RELU_TAPE is not None
                or not (self.training and torch.is_grad_enabled()) or not streams.enabled(real, 'dpasses', finetuning=self.finetuning)):
            return
        if real.dim() > 4:
            real = real[:, 0]
        self._conv_sn_layers()
        self.__dict__['_step_packs'] = prepared[0]
        eu, ev, esig = prepared[1]
        embed = SNEmbeddingFn.apply(label, self.embed.weight_orig, eu, ev, esig, self.__dict__.setdefault('_embed_parts', {}))
        with streams.branch(real.device, 7) as b3:
            real_score, real_features = self.pass_inputs(real, embed, sn_states=ahead[2], strict=self._dstrict())
        self.__dict__['_early_real'] = (embed, b3, real_score, real_features)

    def _dstrict(self):
        return dpass_prec() if (self.training and torch.is_grad_enabled() and dpass_prec()[0] != default_prec()) else None

    def forward(self, data_dict):
        fake, real, label = data_dict['fake_rgbs'], data_dict['target_rgbs'], data_dict['label']
        if fake.dim() > 4:
            fake = fake[:, 0]
        if real.dim() > 4:
            real = real[:, 0]
        prepared = self.__dict__.pop('_prepared', None)
        if prepared is None or not (self.training and torch.is_grad_enabled()):
            prepared = None
            self.__dict__.pop('_prepared_passes', None)
        self.__dict__['_step_packs'] = prepared[0] if prepared is not None else self._fresh_packs()   # new step: the optimizer has changed W_orig
        # label embedding: power iteration on the (98000 x 512 | 1 x 512) matrix by the batched SN kernels, then a row gather scaled
        # by 1/sigma -- W/sigma is never materialised and the backward is row-sparse + rank-1 (SNEmbeddingFn)
        if not label.is_cuda:
            raise RuntimeError('the discriminator runs on the MI355X HIP path only (no CPU fallback)')
        early_real = self.__dict__.pop('_early_real', None)
        if early_real is not None:
            embed = early_real[0]
        else:
            eu, ev, esig = prepared[1] if prepared is not None else self._embed_batch().update(self.training)[0]
            embed = SNEmbeddingFn.apply(label, self.embed.weight_orig, eu, ev, esig, self.__dict__.setdefault('_embed_parts', {}))
        # Pass 1 feeds only generator-side losses; the gradients it would deposit on the discriminator's parameters are erased
        # by optimizer_D.zero_grad() before loss_D.backward (runners/holycow.py:246-248) and no optimizer reads them, so they
        # are not computed unless ``keep_reference_waste`` asks for the reference's exact .grad side effects (parity tests).
        track1 = bool(getattr(self, 'keep_reference_waste', False))
        gstrict = gpass_prec() if (self.training and torch.is_grad_enabled() and gpass_prec()[0] != default_prec()) else None
        dstrict = self._dstrict()
        from latent_pose_reenactment_amd import streams
        if self.training and torch.is_grad_enabled() and streams.enabled(fake, 'dpasses', finetuning=self.finetuning):
            # the three passes are independent given the images and the label embedding: the two discriminator-side passes run on side
            # streams beside the generator-side one, and -- autograd keeps a node on its forward stream -- their backward passes, the whole
            # of loss_D.backward, run beside each other.  The power iterations keep their order: pass k gets the k-th iteration.
            self._conv_sn_layers()
            ahead = self.__dict__.pop('_prepared_passes', None)
            sts = ahead if ahead else [self._sn_batch.update(True) for _ in range(3)]
            # (issued in the reference's order -- the debug tape of the parity tests records ReLU sites in issue order -- from one fork point)
            here = streams.fork_point(fake.device)
            fake_score_G, fake_features = self.pass_inputs(fake, embed if track1 else embed.detach(), track_weights=track1, sn_states=sts[0], strict=gstrict)
            # (this pass and the real-image pass deposit gradients on the same parameters from two streams: this one accumulates into the
            #  parameters' SECOND buffers -- nn.alt_accumulation)
            with streams.branch(fake.device, 6, after=here) as b2, lpnn.alt_accumulation():
                fake_score_D, _ = self.pass_inputs(fake.detach(), embed.detach(), sn_states=sts[1], strict=dstrict)
            if early_real is not None:
                _, b3, real_score, real_features = early_real
            else:
                with streams.branch(fake.device, 7, after=here) as b3:
                    real_score, real_features = self.pass_inputs(real, embed, sn_states=sts[2], strict=dstrict)
            b2.join(fake_score_D)
            b3.join((real_score, real_features))
        else:
            fake_score_G, fake_features = self.pass_inputs(fake, embed if track1 else embed.detach(), track_weights=track1, strict=gstrict)
            fake_score_D, _ = self.pass_inputs(fake.detach(), embed.detach(), strict=dstrict)
            real_score, real_features = self.pass_inputs(real, embed, strict=dstrict)
        data_dict.update(fake_features=fake_features, real_features=real_features, real_embedding=embed,
                         fake_score_G=fake_score_G, fake_score_D=fake_score_D, real_score=real_score)
