"""ResNeXt-50 32x4d identity encoder on the gfx950 kernels: forward AND backward as ONE ``torch.autograd.Function`` -- the reference's
``torchvision.models.resnext50_32x4d(num_classes=512)`` call (embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:26,37-54),
trained in meta-training (runners/holycow.py:34-41 puts the embedder's parameters into optimizer_G).

Data flow (NHWC fp32 activations, 16-bit operand planes for every contraction, as in the generator):
  stem     7x7/2 conv = lp_im2col_planes (147 taps per output pixel as one operand row) + the 1x1 contraction kernel;
           BatchNorm + ReLU + MaxPool(3,2,1) in one pass (lp_bn_relu_maxpool_fwd, 1-byte argmax for the backward)
  block    conv1 1x1 -> BN -> ReLU -> grouped 3x3 (stride 1|2) -> BN -> ReLU -> conv3 1x1 -> BN -> (+ identity | BN(downsample 1x1)) -> ReLU
           1x1 convs:   lp_conv16_fwd / lp_conv16_wgrad on the pixels flattened to one "image" (a 1x1 conv ignores the spatial structure)
           grouped 3x3: lp_gconv16_fwd / lp_gconv16_wgrad (block-diagonal over aligned 64-channel blocks); stride 2 = full-resolution conv +
                        lp_subsample2, its gradients through lp_zero_stuff2 (3 of the 16 blocks)
           BatchNorm:   train mode: lp_bn_train_stats (batch statistics + running-statistics update) -> per-channel (scale, shift) which
                        lp_act_pack (pro 4) applies with the ReLU while it writes the next conv's operand planes; the block output's
                        BN + residual + ReLU is lp_bn_add_act.  Backward: lp_norm_act_bwd (the AdaIN backward kernels with one "image").
  head     AdaptiveAvgPool2d(1) (lp_spatial_mean) + the classifier as a 1x1 contraction over the N frames.
Raw conv outputs: fp32 in the strict (bf16x3) and bf16 modes.  In the fp16 mode they are 16-BIT RESIDENT ("y16", LP_E_Y16=0 turns it
off): the conv epilogue leaves the BatchNorm partials (taken from its fp32 accumulators) and the unscaled fp16 plane of y -- no fp32 y --;
lp_bn_act16 / lp_bn_add_act16 / lp_bn_bwd16_h read 2 B per element where lp_act_pack / lp_bn_add_act / lp_bn_bwd16 read 4, and the ReLU
pattern behind the residual add is read from the block output's operand planes instead of its fp32 copy.  (The stem, the three stride-2
grouped convs and the downsample branches keep fp32 y.)  The saved operand planes are what the weight gradients multiply.  Nothing here
is a torch op except views and allocation."""
import os

import torch

from latent_pose_reenactment_amd import hipops as ops
from latent_pose_reenactment_amd import streams as _streams
from latent_pose_reenactment_amd._lib import PREC_BF16, PREC_BF16X3, PREC_F16


class _BN:
    """per-layer BatchNorm state of one pass: (mean, rstd, scale, shift) [C]"""
    __slots__ = ('mean', 'rstd', 'scale', 'shift')

    def __init__(self, mean, rstd, scale, shift):
        self.mean, self.rstd, self.scale, self.shift = mean, rstd, scale, shift


def supported(n: int, h: int, w: int) -> bool:
    """geometry the kernels cover: the last stage's grouped 3x3 convs (maps of H/32 x W/32) need >= 4 wide maps (weight-gradient tiles
    are >= 4 pixels wide), the classifier contraction N % 4 == 0 and N >= 8.  The 256 x 256 x (B x 8 frames) workload and the 128 x 128
    test size qualify; anything else raises (backbones.ResNeXt.forward: one backend)."""
    return h % 32 == 0 and w % 32 == 0 and h >= 128 and w >= 128 and n % 4 == 0 and n >= 8


Y16 = os.environ.get('LP_E_Y16', '1') != '0'        # fp16 mode: conv outputs stay 16-bit resident
RES16 = os.environ.get('LP_E_RES16', '1') != '0'    # identity shortcuts of bf16 / bf16x3 blocks read the block input from its operand planes; no fp32 block output there
MASK16 = os.environ.get('LP_E_MASK16', '1') != '0'  # block-output ReLU pattern from the operand planes in every mode (0: from the fp32 copy outside the fp16 mode)


def _side_ok(t):
    return _streams.enabled(t, 'wgrad')


def _v16(a, shape):
    """operand planes viewed with another (same-size) leading shape"""
    return ops.Act16(a.hi.view(shape), None if a.lo is None else a.lo.view(shape), a.c, a.inv)


def _conv1x1(a16, pack, prec, bias=None, res=None, amax=False, stats=False, y16=False):
    """1x1 contraction on flattened pixels: a16 [..., C8] planes -> y [P, Cout] fp32 (``stats``: -> (y, ConvStats | None): the BatchNorm
    partials of y over all P pixels, written by the conv epilogue).  ``y16``: y is returned as its fp16 plane (``Act16``), no fp32 y."""
    fa = ops.flat16(a16)
    _, h, w, _ = fa.hi.shape
    if res is not None:
        res = res.view(1, h, w, -1)
    if y16:
        out = ops.conv16(fa, pack, ksize=1, bias=bias, res=res, prec=prec, stats=stats, want_y=False, out16=0, kind='conv1x1')
        return out[1:] if stats else out[1]
    return ops.conv16(fa, pack, ksize=1, bias=bias, res=res, prec=prec, amax=amax, stats=stats, kind='conv1x1')


def _yview(y, shape):
    return _v16(y, shape) if isinstance(y, ops.Act16) else y.view(shape)


def _bn_relu_planes(y, st, prec):
    """operand planes of relu(BatchNorm(y)) for the next conv"""
    if isinstance(y, ops.Act16):
        return ops.bn_act16(y, st.scale, st.shift)
    return ops.act_pack(y, pro=4, scale=st.scale, shift=st.shift, prec=prec)


def bn_state(y, st, gamma, beta, m, counters):
    """train-mode BatchNorm of a conv output: from the partials the conv epilogue left (``st``: no pass over y), else lp_bn_train_stats"""
    if m.track_running_stats:
        counters.append(m.num_batches_tracked)
    rm, rv = (m.running_mean, m.running_var) if m.track_running_stats else (None, None)
    if st is not None:
        return _BN(*ops.norm_stats_finalize(st, 1, gamma.shape[0], gamma, beta, m.eps, running_mean=rm, running_var=rv, momentum=m.momentum))
    if isinstance(y, ops.Act16):          # (a geometry the fused statistics do not cover: none of the supported() sizes)
        y = ops.y16_to_f32(y)
    return _BN(*ops.bn_train_stats(y, gamma, beta, rm, rv, m.momentum, m.eps))


def _wgrad1x1(a16, d16, prec, bias_grad=False):
    return ops.conv_wgrad16(ops.flat16(a16), ops.flat16(d16), ksize=1, prec=prec, bias_grad=bias_grad, kind='wgrad1x1')


class ResNeXtFunction(torch.autograd.Function):
    """inputs: the module (structure, buffers), frames x [N,3,H,W] NCHW fp32, then the parameters in ``net.parameters()`` order.
    output: logits [N, num_classes]."""

    @staticmethod
    def forward(ctx, net, x, *params):
        prec = net.prec
        train = net.training
        need_grad = any(ctx.needs_input_grad[2:])
        if ctx.needs_input_grad[1]:          #  the image gradient is not produced: say so instead of returning None silently
            raise RuntimeError('the HIP encoder does not differentiate with respect to its input frames (detach them)')
        par = dict(zip(net._hip_param_names, params))
        packs = net._hip_packs(par, need_grad)
        bn_eval = None if train else net._hip_eval_affines(par)
        counters = []

        def bn(y, name, st=None):
            if not train:
                return bn_eval[name]
            return bn_state(y, st, par[name + '.weight'].detach(), par[name + '.bias'].detach(), net._hip_bn[name], counters)

        n, _, hin, win = x.shape
        x = x.detach().contiguous()
        if x.dtype not in (torch.float32, torch.float64):      # (fp64 only reaches the CPU emulation of the tests)
            x = x.float()
        # ---- stem
        cols = ops.im2col_planes(x, 7, 2, 3, prec)                               # [N, H/2, W/2, 152]
        h0, w0 = cols.hi.shape[1], cols.hi.shape[2]
        y0, cs = _conv1x1(cols, packs['conv1.weight'][0], prec, stats=True)
        y0 = y0.view(n, h0, w0, 64)
        st0 = bn(y0, 'bn1', cs)
        bprecs = net.block_precs()          # operand mode per bottleneck (a bf16x3 head of the network, an fp16 tail: backbones.ResNeXt.block_precs)
        lprecs = net.layer_precs()          # mode per contraction (the block's mode unless LP_E_HEAD_F16 moves a layer kind of the head to fp16)
        out, out16, idx = ops.bn_relu_maxpool(y0, st0.scale, st0.shift, lprecs[(net._hip_blocks[0][0], 'conv1')], want_idx=need_grad)
        saved_blocks = []
        # ---- bottleneck blocks.  Inside the loop ``prec`` is the block's own mode; the block output's operand planes are written in the
        # mode of the block that consumes them.
        base_prec = prec
        nblk = len(net._hip_blocks)

        def next_prec(b):          # mode of the planes block b writes for its consumer
            return lprecs[(net._hip_blocks[b + 1][0], 'conv1')] if b + 1 < nblk else bprecs[b]
        # identity shortcuts read from the operand planes of the block input (RES16): blocks whose input and output planes share a bf16-class mode
        planes_res = [RES16 and MASK16 and not blk[5] and lprecs[(blk[0], 'conv1')] in (PREC_BF16, PREC_BF16X3) and next_prec(b) == lprecs[(blk[0], 'conv1')]
                      for b, blk in enumerate(net._hip_blocks)]
        for bi, (bname, cin, width, cout, stride, down) in enumerate(net._hip_blocks):
            prec = bprecs[bi]
            p1, p2, p3 = lprecs[(bname, 'conv1')], lprecs[(bname, 'conv2')], lprecs[(bname, 'conv3')]
            nprec = lprecs[(net._hip_blocks[bi + 1][0], 'conv1')] if bi + 1 < len(bprecs) else prec
            y16 = Y16 and p1 == p2 == p3 == PREC_F16
            assert not (y16 and nprec != PREC_F16), 'an fp16 block must be followed by fp16 blocks (lp_bn_add_act16 writes fp16 planes)'
            xin, xin16 = out, out16
            _, h, w, _ = xin16.hi.shape
            y1, cs = _conv1x1(xin16, packs[bname + '.conv1.weight'][0], p1, stats=True, y16=y16)
            y1 = _yview(y1, (n, h, w, width))
            st1 = bn(y1, bname + '.bn1', cs)
            a1 = _bn_relu_planes(y1, st1, p2)
            if stride == 1:
                if y16:
                    _, y2, cs = ops.gconv16(a1, packs[bname + '.conv2.weight'][0], prec=p2, stats=True, want_y=False, out16=True)
                else:
                    y2, cs = ops.gconv16(a1, packs[bname + '.conv2.weight'][0], prec=p2, stats=True)
                ho, wo = h, w
            else:       # (the statistics of the strided output: one pass over the quarter-size tensor)
                y2, cs = ops.subsample2(ops.gconv16(a1, packs[bname + '.conv2.weight'][0], prec=p2)), None
                ho, wo = y2.shape[1], y2.shape[2]
            st2 = bn(y2, bname + '.bn2', cs)
            a2 = _bn_relu_planes(y2, st2, p3)
            y3, cs = _conv1x1(a2, packs[bname + '.conv3.weight'][0], p3, stats=True, y16=y16)
            y3 = _yview(y3, (n, ho, wo, cout))
            st3 = bn(y3, bname + '.bn3', cs)
            xd16 = yd = std = None
            if down:
                xd16 = ops.subsample2_16(xin16) if stride == 2 else xin16
                yd, cs = _conv1x1(xd16, packs[bname + '.downsample.0.weight'][0], p1, stats=True)
                yd = yd.view(n, ho, wo, cout)
                std = bn(yd, bname + '.downsample.1', cs)
                need_f32 = bi + 1 >= nblk or not (net._hip_blocks[bi + 1][5] or planes_res[bi + 1])          # (as below: who reads an fp32 block output)
                out, out16 = ops.bn_add_act(y3, st3.scale, st3.shift, yd, std.scale, std.shift, relu=True, prec=nprec, want_out=need_f32 or not MASK16)
            elif planes_res[bi]:
                # identity shortcut from the operand planes of the block input (hi + lo: the values conv1 multiplied); the fp32 copy of the block
                # output is only written when something reads it: an identity block that takes the fp32 route, or the pooling head behind the last block
                need_f32 = bi + 1 >= nblk or not (net._hip_blocks[bi + 1][5] or planes_res[bi + 1])
                out, out16 = ops.bn_add_act(y3, st3.scale, st3.shift, xin16, relu=True, prec=nprec, want_out=need_f32)
            else:
                out, out16 = ops.bn_add_act(y3, st3.scale, st3.shift, xin, relu=True, prec=nprec)
            if need_grad:
                # (the ReLU pattern of the block output is read from the HI plane of its operand planes -- 2 B per element in every mode (round 6; the
                #  16-bit-resident mode did so since round 3): relu(out) > 0 <=> its bf16 / fp16 rounding > 0, and the fp32 copy is not kept alive)
                saved_blocks.append((xin16, y1, st1, a1, y2, st2, a2, y3, st3, xd16, yd, std, out16 if MASK16 else (out16 if y16 else out), (h, w, ho, wo), (p1, p2, p3)))
        # ---- head
        prec = base_prec
        _, hl, wl, cl = out.shape
        pooled = ops.spatial_mean(out)                                            # [N, 2048]
        fh, fw = ops.flat_hw(n)
        p16 = ops.act_pack(pooled.view(1, fh, fw, cl), pro=0, prec=prec)
        logits = ops.conv16(p16, packs['fc.weight'][0], ksize=1, bias=par['fc.bias'].detach().contiguous(), prec=prec).view(n, -1)
        if counters:
            torch._foreach_add_(counters, 1)
        if need_grad:
            ctx.params = params
            ctx.net, ctx.par, ctx.packs, ctx.train = net, par, packs, train
            ctx.stem = (cols, y0, st0, idx, (h0, w0))
            ctx.blocks = saved_blocks
            ctx.head = (p16, (hl, wl, cl), (fh, fw))
            ctx.n = n
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        if ctx.blocks is None:
            raise RuntimeError('backward through the HIP encoder a second time: its saved activations were released by the first pass (retain_graph is not supported)')
        net, par, packs, train = ctx.net, ctx.par, ctx.packs, ctx.train
        prec = net.prec
        frozen = not train
        n = ctx.n
        grads = {}

        def bn_bwd(dA, y, st, name, prec=None, **kw):
            prec = net.prec if prec is None else prec
            # BatchNorm (+ ReLU) backward straight to the operand planes of dy: the only consumers are the two gradient contractions
            d16, dg, db, g = ops.bn_bwd16(dA, y, par[name + '.weight'].detach(), st.mean, st.rstd, st.scale, st.shift, prec=prec, frozen=frozen, **kw)
            grads[name + '.weight'], grads[name + '.bias'] = dg, db
            return d16, g

        # LP_OVERLAP_WGRAD=1 (off by default: measured 38.8 -> 41.7 ms per meta-training step, profiles/r03_stream_overlap.txt): weight gradients
        # are off the critical path (only the optimizer reads them), so they can be issued on a side stream behind the data-gradient /
        # BatchNorm-backward chain that produces their operands and joined once, at the end.  Their operand planes are then kept alive until
        # that join (a tensor the main stream frees may be re-used by its next allocation while the side stream still reads it).
        wside = _side_ok(d_logits)
        keep = []

        def wgrad(key, fn, *operands):
            if wside:
                keep.extend(operands)
                with _streams.branch(d_logits.device, 4):
                    grads[key] = fn(*operands)
            else:
                grads[key] = fn(*operands)

        # ---- head
        p16, (hl, wl, cl), (fh, fw) = ctx.head
        dl16 = ops.act_pack(d_logits.contiguous().view(1, fh, fw, -1), prec=prec, grad=True)
        dw, db = ops.conv_wgrad16(p16, dl16, ksize=1, prec=prec, bias_grad=True)
        grads['fc.weight'], grads['fc.bias'] = dw.view(par['fc.weight'].shape), db
        d_pooled = ops.conv16(dl16, packs['fc.weight'][1], ksize=1, prec=prec).view(n, cl)
        d_out = ops.spatial_mean_bwd(d_pooled, hl, wl)                            # [N, hl, wl, 2048]
        # ---- blocks, last to first
        for (bname, cin, width, cout, stride, down), sv in zip(reversed(net._hip_blocks), reversed(ctx.blocks)):
            xin16, y1, st1, a1, y2, st2, a2, y3, st3, xd16, yd, std, out, (h, w, ho, wo), (p1, p2, p3) = sv       # operand modes of conv1 (+ downsample), conv2, conv3
            # out = relu(bn3(y3) + skip): g = d_out * [out > 0] reaches bn3 and the skip branch alike
            d16, g = bn_bwd(d_out, y3, st3, bname + '.bn3', p3, mask_mode=2, mask_src=out, want_g=not down)
            wgrad(bname + '.conv3.weight', lambda a, d, k=bname + '.conv3.weight', p=p3: _wgrad1x1(a, d, p).view(par[k].shape), a2, d16)
            dA2 = _conv1x1(d16, packs[bname + '.conv3.weight'][1], p3).view(n, ho, wo, width)
            d16, _ = bn_bwd(dA2, y2, st2, bname + '.bn2', p2)
            if stride == 2:
                d16 = ops.zero_stuff2_16(d16, h, w)                               # adjoint of the subsample of the full-resolution conv
            cg = par[bname + '.conv2.weight'].shape[1]
            wgrad(bname + '.conv2.weight', lambda a, d, cg=cg, p=p2: ops.gconv_wgrad16(a, d, cg, prec=p), a1, d16)
            dA1 = ops.gconv16(d16, packs[bname + '.conv2.weight'][1], prec=p2)
            d16, _ = bn_bwd(dA1, y1, st1, bname + '.bn1', p1)
            wgrad(bname + '.conv1.weight', lambda a, d, k=bname + '.conv1.weight', p=p1: _wgrad1x1(a, d, p).view(par[k].shape), xin16, d16)
            if down:
                dd16, _ = bn_bwd(d_out, yd, std, bname + '.downsample.1', p1, mask_mode=2, mask_src=out)      # same ReLU pattern as bn3: out > 0
                wgrad(bname + '.downsample.0.weight', lambda a, d, k=bname + '.downsample.0.weight', p=p1: _wgrad1x1(a, d, p).view(par[k].shape), xd16, dd16)
                d_xd = _conv1x1(dd16, packs[bname + '.downsample.0.weight'][1], p1)          # [P', cin]
                if stride == 2:
                    d_xin = _conv1x1(d16, packs[bname + '.conv1.weight'][1], p1).view(n, h, w, cin)
                    ops.add_strided2(d_xin, d_xd.view(n, ho, wo, cin))
                else:
                    d_xin = _conv1x1(d16, packs[bname + '.conv1.weight'][1], p1, res=d_xd).view(n, h, w, cin)
            else:
                d_xin = _conv1x1(d16, packs[bname + '.conv1.weight'][1], p1, res=g).view(n, h, w, cin)
            d_out = d_xin
        # ---- stem
        prec = net.prec
        cols, y0, st0, idx, (h0, w0) = ctx.stem
        dA0 = ops.maxpool_bwd(d_out, idx, h0, w0)
        d16, _ = bn_bwd(dA0, y0, st0, 'bn1')
        wgrad('conv1.weight', lambda a, d: _wgrad1x1(a, d, prec).view(par['conv1.weight'].shape), cols, d16)
        if wside:
            torch.cuda.current_stream(d_logits.device).wait_stream(_streams.side_stream(d_logits.device, 4))
        keep.clear()
        ctx.blocks = ctx.stem = ctx.head = None
        from latent_pose_reenactment_amd.nn import fused_accumulate
        return (None, None) + tuple(fused_accumulate(ctx.params, [grads.get(k) for k in net._hip_param_names]))
