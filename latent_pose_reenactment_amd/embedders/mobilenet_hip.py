"""MobileNetV2 pose encoder with autograd on (meta-training trains it: runners/holycow.py:34-41 puts the embedder's parameters into
optimizer_G) -- forward AND backward of ``torchvision.models.mobilenet_v2(num_classes=256)``
(embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:28,56-58) as one ``torch.autograd.Function`` over the gfx950 kernels.
(The no-grad calls -- the fine-tuning step, drive.py -- keep the fused fp32 forward of backbones.MobileNetV2._forward_hip.)

  stem        3x3/2 conv = lp_im2col_planes (27 taps per output pixel as one operand row) + the 1x1 contraction kernel
  1x1 convs   expand / project / features[18]: lp_conv16_fwd (forward, data gradient) and lp_conv16_wgrad on flattened pixels
  depthwise   lp_dwconv3x3_fwd on the RAW expand output with that layer's BatchNorm + ReLU6 applied while loading;
              backward lp_dwconv3x3_dgrad / lp_dwconv3x3_wgrad (fp32, bandwidth-bound)
  BatchNorm   train: lp_bn_train_stats -> (scale, shift); ReLU6 layers: applied by lp_act_pack (pro 3) / the depthwise load;
              the linear bottleneck output (+ residual): lp_bn_add_act;  backward: lp_norm_act_bwd (act_hi = 6 | no activation)
  head        lp_affine_relu6_mean (BN + ReLU6 + global average pool); Dropout stays a torch op between the two Functions;
              classifier = ``LinearRowsFunction`` (1x1 contraction over the N frames)
The contractions of this 0.3 GFLOP-per-frame network are launch- and bandwidth-bound, so they always run in the fp32-class bf16x3
operand mode (fp16 operands through 52 renormalised layers cost 6e-3 on the pose vector)."""
import torch

from latent_pose_reenactment_amd import hipops as ops
from latent_pose_reenactment_amd._lib import PREC_BF16X3
from .resnext_hip import _BN, _conv1x1, _wgrad1x1, bn_state

PREC = PREC_BF16X3


def supported(n: int, h: int, w: int) -> bool:
    """32 | H, W and the last stage's N x H/32 x W/32 pixels a multiple of 4, at least 8 (the 1x1 contractions flatten all pixels of a
    tensor to rows of >= 4).  N = 4 frames of >= 64 x 64 qualify -- the reference's own configs/default.yaml:19-20 gives each GPU 4
    samples -- since round 4 (the classifier pads its N rows, ``LinearRowsFunction``)."""
    last = n * (h // 32) * (w // 32)
    return n >= 1 and h % 32 == 0 and w % 32 == 0 and h >= 32 and w >= 32 and last >= 8 and last % 4 == 0


class MobileNetFeaturesFunction(torch.autograd.Function):
    """inputs: the module, frames x [N,3,H,W], then the parameters of ``net.features`` in ``named_parameters()`` order;
    output: pooled features [N, 1280] (before Dropout and the classifier)."""

    @staticmethod
    def forward(ctx, net, x, *params):
        names = net._hip_feature_param_names
        par = dict(zip(names, params))
        if ctx.needs_input_grad[1]:          #  the image gradient is not produced: say so instead of returning None silently
            raise RuntimeError('the HIP encoder does not differentiate with respect to its input frames (detach them)')
        need_grad = any(ctx.needs_input_grad[2:])
        feats = list(net.features)
        train = feats[0][1].training
        packs = net._hip_train_packs(par, need_grad)
        bn_eval = None if train else net._hip_eval_bn(par)
        counters = []

        def bn(y, name, m, st=None):
            if not train:
                return bn_eval[name]
            return bn_state(y, st, par[name + '.weight'].detach(), par[name + '.bias'].detach(), m, counters)

        n = x.shape[0]
        x = x.detach().contiguous()
        if x.dtype not in (torch.float32, torch.float64):
            x = x.float()
        cols = ops.im2col_planes(x, 3, 2, 1, PREC)                                 # [N, H/2, W/2, 32] (27 taps + pad)
        h0, w0 = cols.hi.shape[1], cols.hi.shape[2]
        c0 = par['0.0.weight'].shape[0]
        y0, cs = _conv1x1(cols, packs['0.0.weight'][0], PREC, stats=True)
        y0 = y0.view(n, h0, w0, c0)
        st0 = bn(y0, '0.1', feats[0][1], cs)
        raw, st_raw = y0, st0              # the tensor the next depthwise conv loads: raw conv output + its BN (ReLU6 applied on load)
        xact = x16 = None                  # block input (fp32, post-BN) and its operand planes
        saved = []
        for bi, blk in enumerate(feats[1:-1], start=1):
            layers = list(blk.conv)
            pre = f'{bi}.conv'
            expand = len(layers) == 4
            rec = {'expand': expand, 'res': blk.use_res, 'x16': x16}
            if expand:
                _, h, w, _ = xact.shape
                hid = layers[0][0].out_channels
                ye, cs = _conv1x1(x16, packs[pre + '.0.0.weight'][0], PREC, stats=True)
                ye = ye.view(n, h, w, hid)
                ste = bn(ye, pre + '.0.1', layers[0][1], cs)
                raw, st_raw = ye, ste
            dwi = 1 if expand else 0
            dw = layers[dwi][0]
            stride = dw.stride[0]
            hin, win = raw.shape[1], raw.shape[2]
            wd = par[f'{pre}.{dwi}.0.weight'].detach().contiguous()
            yd = ops.dwconv3x3(raw, wd, stride, st_raw.scale, st_raw.shift)
            std = bn(yd, f'{pre}.{dwi}.1', layers[dwi][1])
            ad = ops.act_pack(yd, pro=3, scale=std.scale, shift=std.shift, prec=PREC)
            ho, wo = yd.shape[1], yd.shape[2]
            oup = layers[dwi + 1].out_channels
            yp, cs = _conv1x1(ad, packs[f'{pre}.{dwi + 1}.weight'][0], PREC, stats=True)
            yp = yp.view(n, ho, wo, oup)
            stp = bn(yp, f'{pre}.{dwi + 2}', layers[dwi + 2], cs)
            xact, x16 = ops.bn_add_act(yp, stp.scale, stp.shift, xact if blk.use_res else None, relu=False, prec=PREC)
            rec.update(raw=raw, st_raw=st_raw, wd=wd, stride=stride, yd=yd, std=std, ad=ad, yp=yp, stp=stp, pre=pre, dwi=dwi,
                       dims=(hin, win, ho, wo))
            if need_grad:
                saved.append(rec)
        last = feats[-1]
        li = len(feats) - 1
        _, hl, wl, _ = xact.shape
        yl, cs = _conv1x1(x16, packs[f'{li}.0.weight'][0], PREC, stats=True)
        yl = yl.view(n, hl, wl, last[0].out_channels)
        stl = bn(yl, f'{li}.1', last[1], cs)
        pooled = ops.affine_relu6_mean(yl, stl.scale, stl.shift)
        if counters:
            torch._foreach_add_(counters, 1)
        if need_grad:
            ctx.params = params
            ctx.net, ctx.par, ctx.packs, ctx.train, ctx.n = net, par, packs, train, n
            ctx.stem = (cols, y0, st0, (h0, w0))
            ctx.saved = saved
            ctx.last = (x16, yl, stl, li, (hl, wl))
        return pooled

    @staticmethod
    def backward(ctx, d_pooled):
        if getattr(ctx, 'stem', None) is None:
            raise RuntimeError('backward through the HIP encoder a second time: its saved activations were released by the first pass (retain_graph is not supported)')
        net, par, packs, n = ctx.net, ctx.par, ctx.packs, ctx.n
        frozen = not ctx.train
        grads = {}

        def bn_bwd(dA, y, st, name, **kw):
            dx, dg, db, _ = ops.norm_act_bwd(dA, y, par[name + '.weight'].detach(), st.mean, st.rstd, st.scale, st.shift, frozen=frozen, **kw)
            grads[name + '.weight'], grads[name + '.bias'] = dg, db
            return dx

        def bn_bwd16(dA, y, st, name, **kw):          # straight to the operand planes of dy (consumed only by gradient contractions)
            d16, dg, db, _ = ops.bn_bwd16(dA, y, par[name + '.weight'].detach(), st.mean, st.rstd, st.scale, st.shift, prec=PREC, frozen=frozen, **kw)
            grads[name + '.weight'], grads[name + '.bias'] = dg, db
            return d16

        def wshape(k):
            return par[k].shape
        x16, yl, stl, li, (hl, wl) = ctx.last
        dAl = ops.spatial_mean_bwd(d_pooled.contiguous(), hl, wl)
        d16 = bn_bwd16(dAl, yl, stl, f'{li}.1', act_hi=6.0)
        grads[f'{li}.0.weight'] = _wgrad1x1(x16, d16, PREC).view(wshape(f'{li}.0.weight'))
        d_out = _conv1x1(d16, packs[f'{li}.0.weight'][1], PREC).view(n, hl, wl, -1)
        dA_raw = None
        for rec in reversed(ctx.saved):
            pre, dwi = rec['pre'], rec['dwi']
            hin, win, ho, wo = rec['dims']
            # block output = BN(project) (+ block input): a linear BatchNorm
            d16 = bn_bwd16(d_out, rec['yp'], rec['stp'], f'{pre}.{dwi + 2}', mask_mode=1)
            kp = f'{pre}.{dwi + 1}.weight'
            grads[kp] = _wgrad1x1(rec['ad'], d16, PREC).view(wshape(kp))
            dAd = _conv1x1(d16, packs[kp][1], PREC).view(rec['yd'].shape)
            dyd = bn_bwd(dAd, rec['yd'], rec['std'], f'{pre}.{dwi}.1', act_hi=6.0)
            raw, st_raw = rec['raw'], rec['st_raw']
            grads[f'{pre}.{dwi}.0.weight'] = ops.dwconv3x3_wgrad(raw, dyd, rec['stride'], st_raw.scale, st_raw.shift)
            dA_raw = ops.dwconv3x3_dgrad(dyd, rec['wd'], hin, win, rec['stride'])          # w.r.t. relu6(BN(raw))
            if rec['expand']:
                de16 = bn_bwd16(dA_raw, raw, st_raw, f'{pre}.0.1', act_hi=6.0)
                ke = f'{pre}.0.0.weight'
                grads[ke] = _wgrad1x1(rec['x16'], de16, PREC).view(wshape(ke))
                d_out = _conv1x1(de16, packs[ke][1], PREC, res=d_out if rec['res'] else None).view(n, hin, win, -1)
        # the first block has no expand conv: its depthwise input is the stem output
        cols, y0, st0, (h0, w0) = ctx.stem
        d16 = bn_bwd16(dA_raw, y0, st0, '0.1', act_hi=6.0)
        grads['0.0.weight'] = _wgrad1x1(cols, d16, PREC).view(wshape('0.0.weight'))
        ctx.saved = ctx.stem = ctx.last = None
        from latent_pose_reenactment_amd.nn import fused_accumulate
        return (None, None) + tuple(fused_accumulate(ctx.params, [grads.get(k) for k in net._hip_feature_param_names]))


class LinearRowsFunction(torch.autograd.Function):
    """y = x W^T + b over N rows as a 1x1 contraction on the operand-plane kernels (forward, data and weight gradient).  The kernels see
    rows of >= 4 pixels and >= 2 rows: N is padded with zero rows to a multiple of 4, at least 8 (zero rows add nothing to the weight
    gradient; their outputs / gradients are dropped; the bias gradient is taken over the real rows' dy, whose padding is zero too)."""

    @staticmethod
    def forward(ctx, x, w, b):
        n, k = x.shape
        npad = max(8, (n + 3) // 4 * 4)
        xd = x.detach().contiguous()
        if npad != n:
            xd = torch.cat([xd, xd.new_zeros(npad - n, k)])
        fh, fw = ops.flat_hw(npad)
        x16 = ops.act_pack(xd.view(1, fh, fw, k), pro=0, prec=PREC)
        wd = w.detach().contiguous()
        y = ops.conv16(x16, ops.pack_weights(wd, 0, PREC), ksize=1, bias=None if b is None else b.detach().contiguous(), prec=PREC)
        ctx.x16, ctx.wd, ctx.dims = x16, wd, (n, npad, k, fh, fw)
        return y.view(npad, -1)[:n]

    @staticmethod
    def backward(ctx, dy):
        n, npad, k, fh, fw = ctx.dims
        dy = dy.contiguous()
        if npad != n:
            dy = torch.cat([dy, dy.new_zeros(npad - n, dy.shape[1])])
        d16 = ops.act_pack(dy.view(1, fh, fw, -1), prec=PREC, grad=True)
        dx = dw = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw, db = ops.conv_wgrad16(ctx.x16, d16, ksize=1, prec=PREC, bias_grad=True)
            dw = dw.view(ctx.wd.shape)
        if ctx.needs_input_grad[0]:
            dx = ops.conv16(d16, ops.pack_weights(ctx.wd, 1, PREC), ksize=1, prec=PREC).view(npad, k)[:n]
        return dx, dw, (db if ctx.needs_input_grad[2] else None)
