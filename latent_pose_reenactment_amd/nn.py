"""nn.Modules of the hot path, backed by the gfx950 kernels in liblp_hip.so.

The module tree reproduces the reference's ``state_dict`` keys and ``parameters()`` order exactly, so
reference checkpoints and optimizer states drop in:
  * generator  -- generators/vector_pose_unsupervised_segmentation_noBottleneck.py:40-181 (+ generators/common/blocks.py)
The arithmetic below the Module boundary is NOT torch: the whole decoder (17 AdaIN+ReLU prologues, 23 convs, upsampling,
residual adds, tanh/compose head) runs as one ``torch.autograd.Function`` whose forward and backward are sequences of
C-ABI kernel launches on the current stream.  There is no CPU or eager fallback: without a GPU / the .so it raises.
"""
from __future__ import annotations

import math
import os
import weakref
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import hipops as ops
from ._lib import PREC_BF16, PREC_BF16X3, PREC_F16

ADAIN_EPS = 1e-4
SN_EPS_CONV = 1e-4
SN_EPS_DEFAULT = 1e-12

PREC_NAMES = {'bf16': PREC_BF16, 'bf16x3': PREC_BF16X3, 'f16': PREC_F16}


def default_prec() -> int:
    """MFMA operand mode (fp32 accumulate; activations, weights and optimizer state stay fp32 in HBM):
      LP_PREC=f16     default: IEEE fp16 operands (2^-12), gradient operands scaled by a power of two from their amax, 1 MFMA per k-step;
                      full-size outputs 1.7e-4 / tie-masked gradients <= 1.8e-3 from the fp32 CPU path (tests/test_full_size_parity.py)
      LP_PREC=bf16x3  strict: operands split hi + lo bf16, 3 MFMAs per k-step, fp32-class (2.5e-6 / 2.7e-5)
      LP_PREC=bf16    plain bf16 operands (1.3e-3: misses the 1e-3 output gate; kept for comparison)"""
    return PREC_NAMES[os.environ.get('LP_PREC', 'f16')]


def generator_prec() -> int:
    """operand mode of the generator's convs: ``LP_PREC_G`` if set, else the assignment default (GENERATOR_DEFAULT under the global fp16 mode,
    the global mode otherwise)"""
    name = os.environ.get('LP_PREC_G')
    if name:
        return PREC_NAMES[name]
    return PREC_NAMES[GENERATOR_DEFAULT] if default_prec() == PREC_F16 else default_prec()


GENERATOR_DEFAULT = 'bf16x3'      # round 6: the 17-layer AdaIN decoder with fp16 operands (11 bits) sits at 1.0 - 1.7e-3 on 26 of its 27 gradient tensors and at
                                  # ~6e-4 on its own pre-tanh activations -- SURVEY 8d gates every parameter gradient at 1e-3 (profiles/r05_parity_gradients_f16.json)


# ----------------------------------------------------------------------------------------------------------------------
# spectral-norm parameter holder with the legacy-hook key names (weight_orig / weight_u / weight_v [+ bias])
# ----------------------------------------------------------------------------------------------------------------------
class SNWeight(nn.Module):
    """Parameter/buffer set of ``torch.nn.utils.spectral_norm(nn.Conv2d|nn.Linear|nn.Embedding)``.

    ``effective_weight()`` follows the legacy hook (torch/nn/utils/spectral_norm.py::compute_weight) as used at
    generators/common/blocks.py:76-88: in train mode one in-place power iteration on (u, v) under no_grad, then
    ``W_orig / (u . W v)`` with u, v constant for autograd."""

    def __init__(self, shape: Tuple[int, ...], bias: bool, eps: float, fan_in: Optional[int] = None):
        super().__init__()
        out_f = shape[0]
        in_flat = int(math.prod(shape[1:]))
        fan_in = fan_in or in_flat
        if bias:   # registered first: the reference's parameter order per layer is [bias, weight_orig]
            bound = 1 / math.sqrt(fan_in)
            self.bias = nn.Parameter(torch.empty(out_f).uniform_(-bound, bound))
        else:
            self.register_parameter('bias', None)
        w = torch.empty(shape)
        nn.init.kaiming_uniform_(w.view(out_f, in_flat), a=math.sqrt(5))
        self.weight_orig = nn.Parameter(w)
        self.register_buffer('weight_u', F.normalize(torch.randn(out_f), dim=0, eps=eps))
        self.register_buffer('weight_v', F.normalize(torch.randn(in_flat), dim=0, eps=eps))
        self.eps = eps

    def effective_weight(self) -> torch.Tensor:
        w = self.weight_orig
        w_mat = w.reshape(w.shape[0], -1)
        u, v = self.weight_u, self.weight_v
        if self.training:
            with torch.no_grad():
                v = F.normalize(torch.mv(w_mat.t(), u), dim=0, eps=self.eps, out=self.weight_v)
                u = F.normalize(torch.mv(w_mat, v), dim=0, eps=self.eps, out=self.weight_u)
                u, v = u.clone(), v.clone()
        sigma = torch.dot(u, torch.mv(w_mat, v))
        return w / sigma


# Debug aid of the tie-masked parity tests: when a list, every ReLU site of the critic / VGG stacks appends its activation pattern
# (bool, NHWC) in execution order while autograd is recording.  None in production.
RELU_TAPE = None


_TAPE_GRAD = [True]      # grad mode of the caller of the autograd.Function that is running (inside Function.forward it is always off)


def tape_relu(pattern_fn, in_function=False):
    if RELU_TAPE is not None and (_TAPE_GRAD[0] if in_function else torch.is_grad_enabled()):
        RELU_TAPE.append(pattern_fn())


_FUSED_ACCUM = [False]


class fused_grad_accumulation:
    """Context for ``loss.backward()`` of the training step: spectrally normalised conv weights whose ``.grad`` already exists
    (the optimizers' flat gradient arena, zeroed by ``zero_grad``) get their gradient ADDED to ``.grad`` by the kernel that
    finishes it (lp_sn_grad_apply), and autograd receives None for them -- one AccumulateGrad ``add`` launch and one pass over the
    gradient less per weight.  The sum is the same ``0 + g`` / ``g1 + g2`` autograd would form.  Off by default (plain autograd
    semantics, e.g. for torch.autograd.grad and for tests that read the returned gradients)."""

    def __enter__(self):
        self.prev = _FUSED_ACCUM[0]
        _FUSED_ACCUM[0] = True
        ops.sn_defer_begin()          # spectral-norm gradient rules of accumulated conv weights: one batched launch at the exit (hipops.sn_defer_end)

    def __exit__(self, *exc):
        _FUSED_ACCUM[0] = self.prev
        from latent_pose_reenactment_amd import streams
        if exc[0] is None:
            #  every branch that accumulated on a side stream is joined HERE, so no caller can read .grad early; then the deferred
            # spectral-norm jobs run (their operands were produced on the joined streams) and the second accumulation buffers
            # (``alt_accumulation``) are folded into .grad
            streams.join_all()
            ops.sn_defer_end()
            flush_alt_accumulation()
        else:
            ops.sn_defer_end(discard=True)
            #  a backward pass that raised half-way (out of memory, a caught-and-skipped step) leaves partial gradients in the second
            # buffers; zero_grad never touches those, so they are DISCARDED here instead of being added into the next successful step's .grad
            try:
                streams.join_all()
            finally:
                discard_alt_accumulation()


# Two backward passes that run CONCURRENTLY on different streams and deposit gradients on the SAME parameters (the critic's fake-image and
# real-image passes of loss_D.backward: streams.py 'dpasses') must not both ``.grad += g`` from their own kernels -- that is a read-modify-write
# race as soon as the two streams really overlap (hipGraph replays; found by tests/test_train_entry_gpu.py when the real pass moved beside the
# generator, round 4).  Functions whose FORWARD ran inside ``alt_accumulation()`` accumulate into a second, persistent buffer per parameter
# instead; ``flush_alt_accumulation`` (called by ``fused_grad_accumulation.__exit__`` after the streams are joined) adds those buffers to
# .grad with one multi-tensor add and clears them with one multi-tensor zero.
_ALT = {'on': False, 'bufs': {}, 'dirty': {}}


class alt_accumulation:
    def __enter__(self):
        self.prev = _ALT['on']
        _ALT['on'] = True

    def __exit__(self, *exc):
        _ALT['on'] = self.prev


def flush_alt_accumulation():
    dirty = _ALT['dirty']
    if not dirty:
        return
    tgt, src = [g for g, _ in dirty.values()], [b for _, b in dirty.values()]
    torch._foreach_add_(tgt, src)
    torch._foreach_zero_(src)
    dirty.clear()


def discard_alt_accumulation():
    dirty = _ALT['dirty']
    if dirty:
        torch._foreach_zero_([b for _, b in dirty.values()])
        dirty.clear()


_FUSED_ACCUM_ENV = os.environ.get('LP_FUSED_ACCUM', '1') != '0'      # 0: plain autograd accumulation everywhere (diagnosis knob)


def _accum_target(w, alt=False):
    """the tensor a gradient-producing kernel may add into: ``w.grad``, or (``alt``: the pass runs beside another one that feeds the same
    parameters) the parameter's second accumulation buffer (zero between steps)"""
    if not _FUSED_ACCUM[0] or not _FUSED_ACCUM_ENV or not (w.is_leaf and w.requires_grad):
        return None
    g = w.grad
    if not (g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.shape == w.shape):
        return None
    if not alt:
        return g
    ent = _ALT['bufs'].get(id(w))
    if ent is None or ent[0]() is not w or ent[1].shape != g.shape or ent[1].device != g.device:
        import weakref
        for k in [k for k, e in _ALT['bufs'].items() if e[0]() is None]:       # (parameters that are gone: release their buffers)
            del _ALT['bufs'][k]
        ent = (weakref.ref(w), torch.zeros_like(g))
        _ALT['bufs'][id(w)] = ent
    _ALT['dirty'][id(w)] = (g, ent[1])
    return ent[1]


def fused_accumulate(params, grads):
    """For autograd.Functions that produce the gradients of MANY parameters at once (the embedder backbones): inside
    ``fused_grad_accumulation`` the gradients of all leaf parameters whose ``.grad`` exists are added to it by ONE multi-tensor add
    (instead of one AccumulateGrad launch per parameter) and autograd receives None for them.  -> the list to return from backward."""
    out = list(grads)
    if not _FUSED_ACCUM[0]:
        return out
    tgt, src = [], []
    for i, (p, g) in enumerate(zip(params, grads)):
        if g is None:
            continue
        t = _accum_target(p)
        if t is not None:
            tgt.append(t)
            src.append(g.reshape(t.shape) if g.shape != t.shape else g)
            out[i] = None
    if tgt:
        torch._foreach_add_(tgt, src)
    return out


class SNBatch:
    """One-launch spectral normalisation of many ``SNWeight`` layers (lp_sn_power_iter): in train mode each layer's (u, v)
    buffers take one power-iteration step in place, and for every layer the call yields ``(u_used, v_used, sig)`` with
    ``sig = [sigma, 1/sigma]`` (device tensors).  ``1/sigma`` becomes the conv kernels' epilogue scale, so ``W / sigma`` is never
    materialised; backward uses the saved (u, v).  Outputs rotate over static buffer sets (a discriminator runs three passes per
    step, each needing its own copies until its backward is done; static addresses keep hipGraph capture valid).  A set is reused
    only when the autograd graph of the pass it served is gone: every grad-mode update hands out a fresh view object of the first
    layer's sigma, which the consumers keep in their saved state, and the set stays reserved while that object is alive; if all
    sets are reserved (more forward passes than ``SETS`` before a backward: gradient accumulation, two discriminator evaluations)
    another set is allocated instead of silently overwriting one."""
    SETS = 4

    def __init__(self, layers):
        self.layers = list(layers)
        self.sets = None
        self.live = []
        self.key = None
        self.next = 0

    def _new_set(self):
        import struct
        from . import _lib
        assert _lib.lib().lp_sn_desc_bytes() == 72
        rb = _lib.lib().lp_sn_row_block()
        dev = self.layers[0].weight_orig.device
        rows = [l.weight_orig.shape[0] for l in self.layers]
        cols = [l.weight_orig[0].numel() for l in self.layers]
        self.max_rows, self.max_cols = max(rows), max(cols)
        sig = torch.zeros(len(self.layers), 2, dtype=torch.float32, device=dev)
        al = lambda n: (n + 3) // 4 * 4                 # 16-byte aligned slices: the mat-vec kernels use 16-B loads
        uo = torch.zeros(sum(al(r) for r in rows), dtype=torch.float32, device=dev)
        vo = torch.zeros(sum(al(c) for c in cols), dtype=torch.float32, device=dev)
        need = [al(((r + rb - 1) // rb) * c + r + (c + 63) // 64) for r, c in zip(rows, cols)]
        scratch = torch.zeros(sum(need), dtype=torch.float32, device=dev)
        blob = bytearray()
        states = []
        ro = co = so = 0
        for i, l in enumerate(self.layers):
            w, u, v = l.weight_orig.data, l.weight_u, l.weight_v
            assert w.is_contiguous() and w.dtype == torch.float32
            ui, vi, si, pi = uo[ro:ro + rows[i]], vo[co:co + cols[i]], sig[i], scratch[so:so + need[i]]
            blob += struct.pack('<QQQQQQQiifi', w.data_ptr(), u.data_ptr(), v.data_ptr(), ui.data_ptr(), vi.data_ptr(), si.data_ptr(),
                                pi.data_ptr(), rows[i], cols[i], float(l.eps), 0)
            states.append((ui, vi, si))
            ro += al(rows[i]); co += al(cols[i]); so += need[i]
        table = torch.frombuffer(blob, dtype=torch.uint8).clone().to(dev)
        return (table, states, (sig, uo, vo, scratch))

    def _build(self):
        self.sets = [self._new_set() for _ in range(self.SETS)]
        self.live = [None] * self.SETS
        self.next = 0

    def update(self, training: bool):
        from . import _lib
        key = tuple((l.weight_orig.data_ptr(), l.weight_u.data_ptr(), l.weight_v.data_ptr()) for l in self.layers)
        if self.sets is None or key != self.key:
            self._build()
            self.key = key
        idx = None
        for _ in range(len(self.sets)):
            cand = self.next
            self.next = (self.next + 1) % len(self.sets)
            ref = self.live[cand]
            if ref is None or ref() is None:
                idx = cand
                break
        if idx is None:                                  # every set still serves a pass whose backward has not run
            self.sets.append(self._new_set())
            self.live.append(None)
            idx = len(self.sets) - 1
            self.next = 0
        table, states, bufs = self.sets[idx]
        _lib.check(_lib.lib().lp_sn_power_iter(table.data_ptr(), len(self.layers), int(training), self.max_rows, self.max_cols,
                                               torch.cuda.current_stream().cuda_stream), 'lp_sn_power_iter')
        if torch.is_grad_enabled():
            u0, v0, _ = states[0]
            token = bufs[0][0]                           # a NEW view object of layer 0's [sigma, 1/sigma]: alive as long as a consumer's
            self.live[idx] = weakref.ref(token)          # saved state is, i.e. until the pass's autograd graph is released
            states = [(u0, v0, token)] + states[1:]
        else:
            self.live[idx] = None
        return states


def _linear_check(x2):
    if not (x2.is_cuda and x2.dtype == torch.float32 and x2.shape[1] <= 1024 and x2.shape[0] >= 1):
        raise RuntimeError(f'lp_linear_fwd/bwd take fp32 CUDA rows of K <= 1024 features; got {tuple(x2.shape)} {x2.dtype} on {x2.device} '
                           '(one backend: there is no library-GEMM path)')


def _pad4(t):
    """rows of K features -> rows of ceil4(K) (zero columns): the lp_linear_* kernels read 16-byte pieces.  Only toy configurations have K % 4 != 0."""
    k = t.shape[-1]
    return t if k % 4 == 0 else F.pad(t, (0, 4 - k % 4))


_LINEAR_ROWS = 64          # rows per lp_linear_* launch (the kernels keep one accumulator row set per wave): larger batches go in chunks


class ProjScoreFn(torch.autograd.Function):
    """critic head (discriminators/no_landmarks.py:100-108): (pooled, dot) = (sum_hw relu(out), <pooled, embed>) -- one launch forward, one backward
    (round 6; relu + sum + mul + sum and their autograd were ~11 launches per pass).  ``embed`` None: pooled only."""

    @staticmethod
    def forward(ctx, out, embed):
        o = out.detach().contiguous()
        e = None if embed is None else embed.detach().contiguous()
        pooled, dot = ops.proj_score_fwd(o, e)
        ctx.save_for_backward(o, e, pooled)
        if dot is None:
            dot = pooled.new_zeros(())
            ctx.mark_non_differentiable(dot)
        return pooled, dot

    @staticmethod
    def backward(ctx, g_pooled, g_dot):
        o, e, pooled = ctx.saved_tensors
        gp = None if g_pooled is None else g_pooled.contiguous()
        gd = None if (g_dot is None or e is None) else g_dot.contiguous()
        d_out, d_embed = ops.proj_score_bwd(o, e, pooled, gp, gd, ctx.needs_input_grad[0], e is not None and ctx.needs_input_grad[1])
        return d_out, d_embed


class ImagePrepFn(torch.autograd.Function):
    """input side of the VGG criterions (criterions/common/perceptual_loss.py:72-80): NCHW image in [-1, 1] -> ((x + 1) / 2 - mean) / std as an NHWC
    tensor, the reference's fp32 operations in the reference's order, one launch each way (round 6: five ATen launches per image batch)"""

    @staticmethod
    def forward(ctx, x, mean, std):
        ctx.std = std
        return ops.image_prep_fwd(x.detach().contiguous(), mean, std)

    @staticmethod
    def backward(ctx, g):
        return ops.image_prep_bwd(g.contiguous(), ctx.std), None, None


class SNLinearFn(torch.autograd.Function):
    """y = (x W_orig^T) / sigma + b for a spectrally normalised nn.Linear whose sigma comes from SNBatch -- or, with ``sig`` None, a plain
    nn.Linear (FSTH_plus's projector).  Both passes are lp_linear_fwd / lp_linear_bwd (small-batch weight streams with 1/sigma and the
    bias fused; the backward reads W once for dx, the raw weight gradient and the bias gradient), followed for a normalised layer by
    lp_sn_grad_apply (legacy-hook rule dW_orig = G/sigma - <G, W_orig>/sigma^2 u v^T, u, v constant).  Batches of more than 64 rows run as
    64-row chunks of the same kernels.  No other backend: unsupported shapes raise."""

    @staticmethod
    def forward(ctx, x, w, b, u, v, sig):
        ctx.save_for_backward(x, w, u, v, sig)
        ctx.w_param = w if (w.requires_grad and w.is_leaf) else None
        ctx.accum_alt = _ALT['on']
        x2 = _pad4(x.reshape(-1, x.shape[-1]).detach()).contiguous()
        _linear_check(x2)
        wd, bd, alpha = _pad4(w.detach()).contiguous(), (None if b is None else b.detach().contiguous()), (None if sig is None else sig[1:])
        ys = [ops.linear_fwd(x2[i:i + _LINEAR_ROWS], wd, bd, alpha) for i in range(0, x2.shape[0], _LINEAR_ROWS)]
        y = ys[0] if len(ys) == 1 else torch.cat(ys)
        return y.reshape(x.shape[:-1] + (w.shape[0],))

    @staticmethod
    def backward(ctx, g):
        x, w, u, v, sig = ctx.saved_tensors
        k = x.shape[-1]
        g2, x2 = g.reshape(-1, g.shape[-1]).contiguous(), _pad4(x.reshape(-1, k).detach()).contiguous()
        wd, alpha = _pad4(w.detach()).contiguous(), (None if sig is None else sig[1:])
        want = ctx.needs_input_grad[:3]
        parts = [ops.linear_bwd(x2[i:i + _LINEAR_ROWS], wd, g2[i:i + _LINEAR_ROWS], alpha, *want) for i in range(0, x2.shape[0], _LINEAR_ROWS)]
        dx, graw, db = parts[0]
        if len(parts) > 1:
            dx = None if dx is None else torch.cat([p_[0] for p_ in parts])
            graw = None if graw is None else torch.stack([p_[1] for p_ in parts]).sum(0)
            db = None if db is None else torch.stack([p_[2] for p_ in parts]).sum(0)
        if k % 4:          # (toy configurations: drop the zero columns again)
            dx = None if dx is None else dx[:, :k].contiguous()
            graw = None if graw is None else graw[:, :k].contiguous()
            wd = w.detach().contiguous()
        if dx is not None:
            dx = dx.reshape(x.shape)
        dw = graw
        if graw is not None and sig is not None:
            dw = ops.sn_grad_apply(graw, wd, u, v, sig, accum=None if ctx.w_param is None else _accum_target(ctx.w_param, ctx.accum_alt))
        return dx, dw, db, None, None, None


def hip_linear(x, w, b):
    """plain ``F.linear`` on the lp_linear_* kernels (no spectral norm)"""
    return SNLinearFn.apply(x, w, b, None, None, None)


class SNEmbeddingFn(torch.autograd.Function):
    """rows = W_orig[label] / sigma of the spectrally normalised label embedding (discriminators/no_landmarks.py:84-86,152), sigma
    from SNBatch (lp_sn_power_iter): neither W/sigma (98000 x 512 = 200 MB per step in the reference) nor F.embedding's dense
    backward is materialised.  Gradient w.r.t. W_orig (legacy-hook rule, u and v constant):
        G/sigma - (<G, W_orig>/sigma^2) u v^T,   G = scatter of d_rows at ``label``  (B non-zero rows)
    i.e. B sparse rows plus ONE dense rank-1 term with a scalar coefficient.  With fused accumulation both parts are added straight
    into the parameter's ``.grad`` (a rank-1 GEMM update and an index_add).  ``holder['parts']`` receives (label, rows of G/sigma,
    coefficient, u, v): all a data-parallel peer needs to rebuild this rank's gradient -- parallel.GradReducer exchanges those
    B x 513 + 1 numbers instead of all-reducing the dense 200 MB gradient."""

    @staticmethod
    def forward(ctx, label, w, u, v, sig, holder):
        ctx.save_for_backward(label, w, u, v, sig)
        ctx.holder = holder
        ctx.w_param = w if (w.requires_grad and w.is_leaf) else None
        assert not _ALT['on'], 'the label embedding is differentiated by ONE pass (its 200 MB gradient has no second buffer)'
        return w.detach().index_select(0, label) * sig[1]

    @staticmethod
    def backward(ctx, d_rows):
        label, w, u, v, sig = ctx.saved_tensors
        alpha = sig[1]
        g_rows = d_rows * alpha
        coef = (d_rows * w.detach().index_select(0, label)).sum() * (alpha * alpha)
        if ctx.holder is not None:
            ctx.holder['parts'] = (label, g_rows, coef, u, v)
        target = None if ctx.w_param is None else _accum_target(ctx.w_param)
        out = target if target is not None else torch.zeros_like(w)
        if out.is_cuda and out.shape[1] % 4 == 0:
            # lp_sn_embed_grad: the rank-1 term (one read + one write of the gradient) and the B rows, in order -- no library GEMM
            from . import _lib
            _lib.check(_lib.lib().lp_sn_embed_grad(out.data_ptr(), u.contiguous().data_ptr(), v.contiguous().data_ptr(), coef.reshape(1).contiguous().data_ptr(),
                                                   label.to(torch.int64).contiguous().data_ptr(), g_rows.contiguous().data_ptr(), out.shape[0], out.shape[1],
                                                   label.numel(), torch.cuda.current_stream().cuda_stream), 'lp_sn_embed_grad')
        else:
            out.addmm_((u * (-coef))[:, None], v[None, :])
            out.index_add_(0, label, g_rows)
        return None, (None if target is not None else out), None, None, None, None


class _Indexed(nn.Module):
    """Container whose children are named by explicit integer positions (mirrors the sparse indices that
    nn.Sequential gives parameter-less layers in the reference: e.g. ``block.3`` / ``block.7``)."""

    def __init__(self, **children: nn.Module):
        super().__init__()
        for k, m in children.items():
            self.add_module(k.lstrip('_'), m)


class _ResBlockParams(nn.Module):
    """Parameters of blocks.ResBlock(norm_layer='adain') -- generators/common/blocks.py:47-103."""

    def __init__(self, cin: int, cout: int, upsample: bool):
        super().__init__()
        i1, i2 = ('4', '8') if upsample else ('3', '7')
        self.i1, self.i2 = i1, i2
        self.block = _Indexed(**{'_' + i1: SNWeight((cout, cin, 3, 3), False, SN_EPS_CONV),
                                 '_' + i2: SNWeight((cout, cout, 3, 3), False, SN_EPS_CONV)})
        self.has_skip = cin != cout or upsample
        if self.has_skip:
            key = '1' if upsample else '0'
            self.skip_key = key
            self.skip = _Indexed(**{'_' + key: SNWeight((cout, cin, 1, 1), True, SN_EPS_CONV)})
        self.cin, self.cout, self.upsample = cin, cout, upsample

    def convs(self):
        c1, c2 = getattr(self.block, self.i1), getattr(self.block, self.i2)
        sk = getattr(self.skip, self.skip_key) if self.has_skip else None
        return c1, c2, sk


class Constant(nn.Module):
    """noBottleneck.py:31-37 (learned 1 x C x s x s input, init ones)."""

    def __init__(self, *shape):
        super().__init__()
        self.constant = nn.Parameter(torch.ones(1, *shape))


def generator_channels(num_channels: int, max_num_channels: int, image_size: int, const_size: int, num_res_blocks: int):
    """(cin, cout, upsample) per decoder block -- noBottleneck.py:60-78."""
    n_up = int(math.log2(image_size / const_size))
    nonclamped = num_channels * (2 ** n_up)
    cur = min(nonclamped, max_num_channels)
    blocks = [(cur, cur, False)] * num_res_blocks
    for _ in range(n_up):
        cin = cur
        nonclamped //= 2
        cur = min(nonclamped, max_num_channels)
        blocks.append((cin, cur, True))
    return blocks


# ----------------------------------------------------------------------------------------------------------------------
# the decoder as ONE autograd Function over HIP kernels
# ----------------------------------------------------------------------------------------------------------------------
# fp16 mode, OPT-IN (LP_G_Y16=1): the decoder's conv outputs on the large maps stay 16-bit resident (round 5).  Measured (profiles/r05_generator_y16.txt):
# generator forward + backward 5.12 -> 5.00 ms, meta-training step -0.1 .. -0.3 ms; outputs 1.64e-4 -> 1.83e-4, tie-masked gradients 1.68e-3 ->
# 1.86e-3 .. 2.16e-3 -- beyond the 2e-3 this package gates the fp16 generator gradients at, for < 1 % of the step: off by default.
G_Y16 = os.environ.get('LP_G_Y16', '0') != '0'
# smallest map (output height) that runs 16-bit resident: 64 -- at 32 x 32 the launches that do not cover the fused statistics would need a decode +
# statistics pass (measured: two extra launch pairs per step), and 6 % of the decoder's activation bytes live there
Y16_MIN_MAP = int(os.environ.get('LP_G_Y16_MIN', '64'))
# x2-upsampled 3x3 convs (conv1 of every up block) in their PHASE-DECOMPOSED forms (round 6; csrc/conv_dma.hip KS = 2): forward per output phase a
# 2 x 2 conv on the low-resolution planes, data gradient ONE launch on the low-resolution grid (no 2H x 2W dA, no 2 x 2 sum in the AdaIN
# backward) -- 4/9 of the matrix work each, exact algebra (the coinciding taps are summed in fp32 inside the weight pack).  bf16x3, N = 8
# (profiles/r06_phase_conv.txt): forward 114 / 160 / 161 / 176 us -> 64 / 83 / 89 / 108 us, data gradient 113 / 184 / 210 / 295 us -> 71 / 112 / 86 / 98 us
# at 32^2 .. 256^2.  LP_G_PHASE=0: the fused-upsample kernels; maps below PHASE_MIN_OUT keep them as well.
PHASE_UP = os.environ.get('LP_G_PHASE', '1') != '0'
PHASE_MIN_OUT = int(os.environ.get('LP_G_PHASE_MIN', '16'))


def phase_conv(up: bool, hout: int, reflect: bool) -> bool:
    return bool(up) and PHASE_UP and hout >= PHASE_MIN_OUT and not reflect


G_F16_TAIL_DEFAULT = 0      # decoder blocks (from the output end) that run fp16 operands in the default (bf16x3) generator assignment
RAW16_SKIP = os.environ.get('LP_G_RAW16', '1') != '0'      # conv2's epilogue also writes the raw planes of the block output for the next skip conv (0: a pack launch)


class _DecoderFunction(torch.autograd.Function):
    """inputs: affine [B, n_aff] (projector output; per AdaIN: C biases then C weights, noBottleneck.py:108-125),
    constant [1,C,s,s], then per block (w1, w2[, w_skip, b_skip]) and (w_head, b_head) -- the W_ORIG parameters themselves: 1/sigma of each
    conv (``cfg['sn']``, from SNBatch) is the ``alpha`` of its launch's epilogue, W/sigma is never formed, and the weight-gradient launches
    return the gradient w.r.t. W_orig through the legacy-hook rule (``snw``).
    outputs: fake_rgbs [B,3,S,S], fake_segm [B,1,S,S] (NCHW, noBottleneck.py:170-181)."""

    @staticmethod
    def forward(ctx, cfg, affine, constant, *weights):
        blocks, prec = cfg['blocks'], cfg['prec']
        precs = cfg.get('precs') or [prec] * len(blocks)      # operand mode per block (``prec``: the head conv's and the default)
        need_grad = cfg['need_grad']
        sn = cfg['sn']                    # per entry of `weights`: (u_used, v_used, [sigma, 1/sigma]) for conv weights, None for biases
        packs = cfg.get('packs')          # forward packs prepared by the module (inference: cached; training: one batched launch)

        def fpack(i, w, phase=False, prec=prec):
            return packs[i] if packs is not None else ops.pack_weights(w.detach().contiguous(), 2 if phase else 0, prec)
        B = affine.shape[0]
        affine = affine.contiguous()
        wl = list(weights)
        x = constant.detach().permute(0, 2, 3, 1).expand(B, -1, -1, -1).contiguous()    # NHWC
        saved = []
        off = 0
        wi = 0

        def aff(c):
            nonlocal off
            beta, gamma = affine[:, off:off + c], affine[:, off + c:off + 2 * c]
            o = off
            off += 2 * c
            return gamma, beta, o

        # 16-BIT-RESIDENT conv outputs (round 5, fp16 mode, maps of >= 64 x 64 -- Y16_MIN_MAP; the conv epilogue also leaves the norm statistics there): a
        # conv between two AdaIN blocks writes the UNSCALED fp16 plane of y (+ the {count, mean, M2} partials from its fp32 accumulators) and no
        # fp32 y.  The plane IS the raw operand of the next block's 1x1 skip conv (exactly what lp_act_pack pro 0 produced from fp32 y), the
        # AdaIN + ReLU prologue reads 2 B per element (lp_adain_act16), and the backward recomputes x-hat / the ReLU pattern from the same plane
        # (lp_adain_relu_bwd16).  Per element over forward + backward: a conv1 output 16 B -> 8 B, a block output 22 B -> 8 B.  LP_G_Y16=0: fp32.
        # gen_padding='reflection' (noBottleneck.py:53-58: nn.ReflectionPad2d(1) in front of the ResBlocks' 3x3 convs, blocks.py:76-88; the head conv
        # keeps its zero padding, noBottleneck.py:80-88): the zero-padded conv + the border correction of csrc/reflect_border.hip; the statistics of a
        # conv output are then taken after the correction (no epilogue partials, no 16-bit-resident outputs)
        reflect = bool(cfg.get('reflect'))
        y16 = bool(cfg.get('y16')) and prec == PREC_F16 and all(p_ == PREC_F16 for p_ in precs) and not reflect

        def dims(t):
            return tuple(t.hi.shape) if isinstance(t, ops.Act16) else tuple(t.shape)

        def in_stats(t, cs, gamma, beta):
            # instance-norm statistics of a conv output: from the {count, mean, M2} partials its epilogue left (no pass over the tensor),
            # else -- maps under 64 pixels, split-K launches, the constant input -- by the two-launch statistics kernel
            if cs is not None:
                return ops.norm_stats_finalize(cs, dims(t)[0], dims(t)[3], gamma, beta, ADAIN_EPS)
            return ops.instnorm_stats(ops.y16_to_f32(t) if isinstance(t, ops.Act16) else t, gamma, beta, ADAIN_EPS)

        def norm_planes(t, st, prec=prec):          # operand planes of relu(AdaIN(t))
            if isinstance(t, ops.Act16):
                return ops.adain_act16(t, st[2], st[3])
            return ops.act_pack(t, pro=1, scale=st[2], shift=st[3], prec=prec)

        def conv_out(a, pk, hout, w_orig, prec, raw16=False, **kw):
            """-> (y fp32 | the fp16 plane of y, statistics partials | None, raw operand planes of y | None).  ``raw16`` (round 6): the epilogue
            also writes the operand planes of y itself -- the input of the NEXT block's 1x1 skip conv (lp_act_pack prologue 0 of y: one launch
            and one read of y less per up block)"""
            if reflect:
                y = ops.conv16(a, pk, prec=prec, **kw)
                return ops.reflect_border_fwd(a, w_orig.detach().contiguous(), kw.get('alpha'), y, prec=prec, upsample=bool(kw.get('upsample'))), None, None
            if y16 and hout >= Y16_MIN_MAP and pk.rows % 8 == 0:
                _, o16, cs = ops.conv16(a, pk, prec=prec, stats=True, want_y=False, out16=0, **kw)
                return o16, cs, None
            if raw16 and pk.rows % 8 == 0:
                y, o16, cs = ops.conv16(a, pk, prec=prec, stats=True, out16=0, **kw)
                return y, cs, o16
            return ops.conv16(a, pk, prec=prec, stats=True, **kw) + (None,)
        x_cs = x_raw16 = None
        for bi_, (cin, cout, up) in enumerate(blocks):
            pb = precs[bi_]
            w1, w2 = wl[wi], wl[wi + 1]
            wi += 2
            has_skip = (cin != cout) or up
            g0, b0, o0 = aff(cin)
            g1, b1, o1 = aff(cout)
            hout = dims(x)[1] * (2 if up else 1)
            # AdaIN + ReLU are applied ONCE per tensor while it is packed to the conv's 16-bit operand planes (the same planes feed
            # the weight gradient in backward); the convs themselves stage their operands by LDS-DMA only
            st0 = in_stats(x, x_cs, g0, b0)
            a0 = norm_planes(x, st0, pb)
            ph1 = phase_conv(up, hout, reflect)
            p1 = fpack(wi - 2, w1, ph1, pb)
            h1, cs1, _ = conv_out(a0, p1, hout, w1, pb, ksize=3, upsample=up, alpha=sn[wi - 2][2][1:], **({'phase': True} if ph1 else {}))
            st1 = in_stats(h1, cs1, g1, b1)
            a1 = norm_planes(h1, st1, pb)
            xs = None
            if has_skip:
                ws, bs = wl[wi], wl[wi + 1]
                wi += 2
                ps = fpack(wi - 2, ws, False, pb)
                xs = x if isinstance(x, ops.Act16) else x_raw16 if x_raw16 is not None else ops.act_pack(x, pro=0, prec=pb)
                s = ops.conv16(xs, ps, ksize=1, bias=bs.detach().contiguous(), alpha=sn[wi - 2][2][1:], prec=pb)   # 1x1 commutes with nearest upsampling
                rs = 1 if up else 0
            else:
                assert not isinstance(x, ops.Act16), 'an identity skip adds the fp32 block input (blocks without a skip conv sit on the smallest maps)'
                s, rs = x, 0
            i2 = wi - (3 if has_skip else 1)
            p2 = fpack(i2, w2, False, pb)
            nxt = blocks[bi_ + 1] if bi_ + 1 < len(blocks) else None
            out, x_cs, x_raw16 = conv_out(a1, p2, hout, w2, pb, raw16=RAW16_SKIP and nxt is not None and (nxt[0] != nxt[1] or nxt[2]) and precs[bi_ + 1] == pb,
                                          ksize=3, res=s, res_shift=rs, alpha=sn[i2][2][1:])
            if need_grad:
                saved.append((x, h1, st0, st1, o0, o1, a0, a1, xs))
            if cfg.get('debug') is not None:      # activation patterns of the AdaIN+ReLU sites (tie-masked parity checks)
                cfg['debug'].setdefault('relu_planes', []).extend([a0.hi, a1.hi])
            x = out
        ch = blocks[-1][1]
        gh, bh, oh = aff(ch)
        sth = in_stats(x, x_cs, gh, bh)
        wh, bhd = wl[wi], wl[wi + 1]
        ph = fpack(wi, wh)
        ah = norm_planes(x, sth)
        z = ops.conv16(ah, ph, ksize=3, bias=bhd.detach().contiguous(), alpha=sn[wi][2][1:], prec=prec)
        if cfg.get('debug') is not None:
            cfg['debug'].setdefault('relu_planes', []).append(ah.hi)
        t, rgbs, segm = ops.head_fwd(z, want_t=need_grad)
        if need_grad:
            ctx.cfg = cfg
            ctx.saved = saved
            ctx.head = (x, sth, oh, t, ah)
            ctx.affine = affine
            ctx.weights = [w.detach() for w in wl]
            ctx.params = wl                  # the parameter tensors themselves: fused accumulation adds into their .grad
            ctx.const_shape = constant.shape
        return rgbs, segm

    @staticmethod
    def backward(ctx, d_rgbs, d_segm):
        cfg = ctx.cfg
        blocks, prec = cfg['blocks'], cfg['prec']
        precs = cfg.get('precs') or [prec] * len(blocks)      # operand mode per block; ``prec``: the head conv
        f16 = prec == PREC_F16          # gradient tensors become fp16 operands: their producers fold max|.| in (no amax pass)
        sn = cfg['sn']
        reflect = bool(cfg.get('reflect'))
        affine, wl = ctx.affine, ctx.weights
        params = ctx.params
        d_affine = torch.zeros_like(affine)

        def border(i, a, dy, dA, up=False, prec=prec):
            """gen_padding='reflection': the border terms of conv weight i -- their share of the weight gradient (through the same spectral-norm
            rule, into the same .grad) and of the data gradient dA (in place); a = the conv's operand planes, dy = the fp32 gradient of its output"""
            corr = ops.reflect_border_wgrad(a, dy, prec=prec, upsample=up, sn=snw(i), accum=_accum_target(params[i]))
            if corr is not None:
                grads[i] = grads[i] + corr
            ops.reflect_border_dgrad(dy, wl[i].contiguous(), sn[i][2][1:], dA)

        def snw(i):          # (W_orig, u, v, sig) of conv weight i -> wgrad returns the gradient w.r.t. W_orig
            return (wl[i],) + tuple(sn[i])
        grads: List[Optional[torch.Tensor]] = [None] * len(wl)

        def slices(o, c):      # (gamma, dgamma, dbeta) views for the AdaIN whose params start at column o
            return affine[:, o + c:o + 2 * c], d_affine[:, o + c:o + 2 * c], d_affine[:, o:o + c]

        x, sth, oh, t, ah = ctx.head
        ch = blocks[-1][1]
        dz = ops.head_bwd(t, d_rgbs.contiguous(), None if d_segm is None else d_segm.contiguous(), amax=f16)
        wi = len(wl) - 2
        if ops.thin_wgrad_supported(ch, dz.shape[3], 3, 1, dz.shape[2]):
            thin = ops.thin_wgrad16 if isinstance(x, ops.Act16) else ops.thin_wgrad
            grads[wi], grads[wi + 1] = thin(x, dz, ksize=3, pro=1, scale=sth[2], shift=sth[3], sn=snw(wi),
                                            accum=_accum_target(params[wi]), bias_grad=True)
        else:
            grads[wi], grads[wi + 1] = ops.conv_wgrad16(ah, ops.act_pack(dz, prec=prec, grad=True), ksize=3, prec=prec, sn=snw(wi),
                                                        accum=_accum_target(params[wi]), bias_grad=True)
        packsT = cfg.get('packsT')        # dgrad packs from the same batched launch (training), else packed on demand

        def tpack(i, small_k=False, phase=False, prec=prec):
            return packsT[i] if packsT is not None else ops.pack_weights(wl[i].contiguous(), 3 if phase else 1, prec, small_k=small_k)
        pT = tpack(wi, small_k=True)
        dA = ops.conv(dz, pT, ksize=3, alpha=sn[wi][2][1:], prec=prec, grad=True)
        g, dg, db = slices(oh, ch)
        # bf16 / bf16x3 (round 6): gradient operands carry no scale there, so the AdaIN backward (and the 2x2 sum of the skip branch) write the
        # operand planes of their result themselves -- lp_adain_relu_bwd_planes / lp_sum2x2_planes: no pack launch, and the conv1-output gradient
        # (dh1), whose only consumers are the two contractions, never exists in fp32
        # A gradient's operand planes are written in the mode of the block that CONSUMES them (``precs``: the blocks of one decoder may differ).
        def is_direct(p_):
            return p_ in (PREC_BF16, PREC_BF16X3)
        dbg = cfg.get('debug')
        dx16 = None
        if is_direct(precs[-1]):
            dx, dx16 = ops.adain_relu_bwd(dA, x, None, g, sth[0], sth[1], sth[2], sth[3], dg, db, False, planes=precs[-1])
        else:
            dx = ops.adain_relu_bwd(dA, x, None, g, sth[0], sth[1], sth[2], sth[3], dg, db, False, amax=precs[-1] == PREC_F16)
        if dbg is not None:
            dbg['dz'] = dz; dbg['dA_head'] = dA; dbg[f'dx{len(blocks)}'] = dx

        for bi in range(len(blocks) - 1, -1, -1):
            cin, cout, up = blocks[bi]
            has_skip = (cin != cout) or up
            x, h1, st0, st1, o0, o1, a0, a1, xs = ctx.saved[bi]
            wi -= 4 if has_skip else 2
            prec = precs[bi]                 # this block's operand mode; the gradient it hands on is packed for block bi - 1
            pnext = precs[bi - 1] if bi > 0 else prec
            direct, f16 = is_direct(prec), prec == PREC_F16
            d_out = dx
            d16 = dx16 if dx16 is not None else ops.act_pack(d_out, prec=prec, grad=True)       # packed once: operand of conv2's weight AND data gradient
            # conv2 (+ AdaIN1/ReLU prologue)
            grads[wi + 1] = ops.conv_wgrad16(a1, d16, ksize=3, prec=prec, sn=snw(wi + 1), accum=_accum_target(params[wi + 1]))
            dA1 = ops.conv16(d16, tpack(wi + 1, prec=prec), ksize=3, alpha=sn[wi + 1][2][1:], prec=prec)
            if reflect:
                border(wi + 1, a1, d_out, dA1, prec=prec)
            g, dg, db = slices(o1, cout)
            if direct:
                dh1, dh16 = ops.adain_relu_bwd(dA1, h1, None, g, st1[0], st1[1], st1[2], st1[3], dg, db, False, planes=prec, keep_dx=reflect or dbg is not None)
            else:
                dh1 = ops.adain_relu_bwd(dA1, h1, None, g, st1[0], st1[1], st1[2], st1[3], dg, db, False, amax=f16)
                dh16 = None
            # skip branch: out += up2(conv1x1(x) + b)
            if has_skip:
                if up:
                    ds16 = ops.sum2x2_planes(d_out, prec) if direct else ops.act_pack(ops.sum2x2(d_out, amax=f16), prec=prec, grad=True)
                else:
                    ds16 = d16
                grads[wi + 2], grads[wi + 3] = ops.conv_wgrad16(xs, ds16, ksize=1, prec=prec, sn=snw(wi + 2),
                                                                accum=_accum_target(params[wi + 2]), bias_grad=True,
                                                                bias_accum=_accum_target(params[wi + 3]))
                dx_skip = ops.conv16(ds16, tpack(wi + 2, prec=prec), ksize=1, alpha=sn[wi + 2][2][1:], prec=prec)
            else:
                dx_skip = d_out
            # conv1 (+ AdaIN0/ReLU/upsample prologue)
            if dh16 is None:
                dh16 = ops.act_pack(dh1, prec=prec, grad=True)
            grads[wi] = ops.conv_wgrad16(a0, dh16, ksize=3, upsample=up, prec=prec, sn=snw(wi), accum=_accum_target(params[wi]))
            ph1 = phase_conv(up, dh16.hi.shape[1], reflect)
            if ph1:      # phase form of the data gradient: the gradient w.r.t. the LOW-resolution AdaIN output from one launch (2 x 2 sum included)
                dA0 = ops.conv16(dh16, tpack(wi, phase=True, prec=prec), ksize=3, alpha=sn[wi][2][1:], prec=prec, phase_dgrad=True)
            else:
                dA0 = ops.conv16(dh16, tpack(wi, prec=prec), ksize=3, alpha=sn[wi][2][1:], prec=prec)
            if reflect:
                border(wi, a0, dh1, dA0, up, prec=prec)
            up_bwd = up and not ph1          # (the fused-upsample form hands the AdaIN backward a 2H x 2W gradient to sum)
            g, dg, db = slices(o0, cin)
            dx16 = None
            if is_direct(pnext) and bi > 0:          # (block 0's input gradient only feeds the learned constant: fp32)
                dx, dx16 = ops.adain_relu_bwd(dA0, x, dx_skip, g, st0[0], st0[1], st0[2], st0[3], dg, db, up_bwd, planes=pnext)
            else:
                dx = ops.adain_relu_bwd(dA0, x, dx_skip, g, st0[0], st0[1], st0[2], st0[3], dg, db, up_bwd, amax=pnext == PREC_F16 and bi > 0)
            if dbg is not None:
                dbg[f'dx{bi}'] = dx; dbg[f'dh1_{bi}'] = dh1; dbg[f'dxskip{bi}'] = dx_skip
        d_const = dx.sum(dim=0, keepdim=True).permute(0, 3, 1, 2).contiguous()
        return (None, d_affine, d_const, *grads)


class Generator(nn.Module):
    """Drop-in for generators/vector_pose_unsupervised_segmentation_noBottleneck.py::Generator (same constructor
    arguments, state_dict keys, forward(data_dict) contract, enable_finetuning)."""

    def __init__(self, padding, in_channels, out_channels, num_channels, max_num_channels, identity_embedding_size,
                 pose_embedding_size, norm_layer, gen_constant_input_size, gen_num_residual_blocks, output_image_size,
                 prec: Optional[int] = None):
        super().__init__()
        if padding not in ('zero', 'reflection'):
            raise Exception('Incorrect `padding` argument, required `zero` or `reflection`')       # (noBottleneck.py:57-58)
        self.reflect = padding == 'reflection'
        if self.reflect and gen_constant_input_size < 4:
            #  nn.ReflectionPad2d(1) accepts maps of >= 2 x 2; the border-correction kernels (csrc/reflect_border.hip) cover >= 4 x 4 --
            # the shipped 4 x 4 constant input.  Say so here instead of failing with LP_ERR_ARG in the first forward.
            raise NotImplementedError('gen_padding=\'reflection\' on the HIP path needs gen_constant_input_size >= 4 (the reflection border kernels cover maps of >= 4 x 4)')
        if 'in' not in norm_layer:
            raise NotImplementedError("only norm_layer='in' is implemented on the HIP path")
        assert math.log2(output_image_size / gen_constant_input_size).is_integer(), \
            "`gen_constant_input_size` must be `image_size` divided by a power of 2"
        if out_channels != 4:
            raise NotImplementedError('the head kernel composes exactly RGB + mask (out_channels + 1 == 4)')
        self.blocks_cfg = generator_channels(num_channels, max_num_channels, output_image_size, gen_constant_input_size,
                                             gen_num_residual_blocks)
        c0 = self.blocks_cfg[0][0]
        self.constant = Constant(c0, gen_constant_input_size, gen_constant_input_size)
        dec = {}
        for i, (cin, cout, up) in enumerate(self.blocks_cfg):
            dec[f'_{i}'] = _ResBlockParams(cin, cout, up)
        nb = len(self.blocks_cfg)
        # indices nb, nb+1 are the head AdaIN and ReLU (no parameters), nb+2 the SN conv, nb+3 the Tanh
        dec[f'_{nb + 2}'] = SNWeight((out_channels, self.blocks_cfg[-1][1], 3, 3), True, SN_EPS_CONV)
        self.decoder_blocks = _Indexed(**dec)
        self.identity_embedding_size = identity_embedding_size
        self.pose_embedding_size = pose_embedding_size
        joint = identity_embedding_size + pose_embedding_size
        self.num_affine_params = sum(2 * (a + b) for a, b, _ in self.blocks_cfg) + 2 * self.blocks_cfg[-1][1]
        self.affine_params_projector = self._build_projector(joint)
        self.finetuning = False
        self.prec = generator_prec() if prec is None else prec
        # forwards that keep NO autograd state (drive.py, the fine-tuning bootstrap, visualisation under no_grad): only the 1e-3 OUTPUT gate applies,
        # which fp16 operands meet with a wide margin (fake_rgbs 1.6e-4 at 256 x 256) at a third of the matrix work -- the default assignment's
        # bf16x3 is for the gradients.  An explicit ``prec`` / LP_PREC_G / a global mode other than f16 applies to every forward.
        self.infer_prec = self.prec if (prec is not None or os.environ.get('LP_PREC_G') or default_prec() != PREC_F16) else PREC_F16
        # the LAST ``f16_tail`` blocks of the decoder (the widest maps: most of its matrix work) may run fp16 operands inside a bf16x3 generator:
        # rounding injected there passes through few layers (LP_G_F16_TAIL; DESIGN section 2 has the measured gradient figures per setting)
        self.f16_tail = int(os.environ.get('LP_G_F16_TAIL', str(G_F16_TAIL_DEFAULT))) if prec is None else 0

    def block_precs(self, prec):
        """operand mode per decoder block for a pass whose base mode is ``prec``"""
        nb = len(self.blocks_cfg)
        k = min(self.f16_tail, nb) if prec == PREC_BF16X3 else 0
        return [prec] * (nb - k) + [PREC_F16] * k

    def _weight_precs(self, precs, prec):
        """operand mode per entry of ``_conv_weights``' list (biases: their conv's), the head conv in the base mode"""
        out = []
        for (cin, cout, up), p_ in zip(self.blocks_cfg, precs):
            out += [p_] * (4 if (cin != cout or up) else 2)
        return out + [prec, prec]

    # ---- the projector is the only part in which the generator plugins differ (noBottleneck.py:96-101 vs FSTH_plus.py:96-103)
    def _build_projector(self, joint):
        hidden = max(joint, 512)
        return _Indexed(_0=SNWeight((hidden, joint), True, SN_EPS_DEFAULT), _2=SNWeight((self.num_affine_params, hidden), True, SN_EPS_DEFAULT))

    def _projector_sn_layers(self):
        return [self.affine_params_projector._modules['0'], self.affine_params_projector._modules['2']]

    def _pose_vector(self, data_dict):
        return data_dict['pose_embedding']

    def _project(self, joint, states):
        """SN-Linear -> ReLU -> SN-Linear on the lp_linear_* weight-stream kernels -- B x 768 x 768 and B x 768 x 13056 -- with the
        1/sigma of the batched power iteration and the legacy-hook weight gradient rule (SNLinearFn)"""
        p0, p2 = self.affine_params_projector._modules['0'], self.affine_params_projector._modules['2']
        h = torch.relu(SNLinearFn.apply(joint, p0.weight_orig, p0.bias, *states[-2]))
        return SNLinearFn.apply(h, p2.weight_orig, p2.bias, *states[-1])

    def get_num_affine_params(self):
        return self.num_affine_params

    def enable_finetuning(self, data_dict=None):
        """noBottleneck.py:139-163: the identity embedding becomes a trainable parameter."""
        if data_dict is None:
            some_parameter = next(iter(self.parameters()))
            identity_embedding = torch.rand(1, self.identity_embedding_size).to(some_parameter)
        else:
            identity_embedding = data_dict['embeds']
        if self.finetuning:
            with torch.no_grad():
                self.identity_embedding.copy_(identity_embedding)
        else:
            self.identity_embedding = nn.Parameter(identity_embedding)
            self.finetuning = True

    def _affine_params(self, data_dict):
        pose = self._pose_vector(data_dict)
        if self.finetuning:
            identity = self.identity_embedding.expand(len(pose), -1)
        else:
            identity = data_dict['embeds']
        joint = torch.cat((identity, pose), dim=1)
        if not joint.is_cuda:
            raise RuntimeError('the generator runs on the MI355X HIP path only (no CPU fallback); move the model and inputs to cuda')
        states = self.__dict__.pop('_prepared_sn', None)         # prepare_step() ran ahead (beside the encoders): use its result
        if states is None or not (self.training and torch.is_grad_enabled()):
            states = self._sn_update()
        self.__dict__['_sn_states'] = states
        return self._project(joint, states)

    def _sn_update(self):
        # ONE launch power-iterates every spectrally normalised layer of the generator (23 convs + 2 projector linears)
        layers, slots = self._sn_layers()
        if self.__dict__.get('_sn_batch') is None or self._sn_batch.layers != layers:
            self.__dict__['_sn_batch'] = SNBatch(layers)
        return self._sn_batch.update(self.training)

    def _conv_weights(self, states):
        """(weights, sn): the decoder's conv weights / biases in _DecoderFunction's argument order, with each conv's SN state"""
        weights, sn = [], []
        nb = len(self.blocks_cfg)
        k = 0
        for i in range(nb):
            c1, c2, sk = self.decoder_blocks._modules[str(i)].convs()
            weights += [c1.weight_orig, c2.weight_orig]
            sn += [states[k], states[k + 1]]
            k += 2
            if sk is not None:
                weights += [sk.weight_orig, sk.bias]
                sn += [states[k], None]
                k += 1
        head = self.decoder_blocks._modules[str(nb + 2)]
        weights += [head.weight_orig, head.bias]
        sn += [states[k], None]
        return weights, sn

    def _train_packs_update(self, weights, sn):
        """training: every conv weight changed in the last optimizer step -> ONE launch re-packs all of them, both orientations"""
        conv_idx = [i for i, s_ in enumerate(sn) if s_ is not None]
        ph = self._phase_weight_indices()          # conv1 of the up blocks that run in their phase forms: pack modes 2 / 3 instead of 0 / 1
        specs = [(weights[i], 2 if i in ph else 0, False) for i in conv_idx] + [(weights[i], 3 if i in ph else 1, i == len(weights) - 2) for i in conv_idx]
        wp = self._weight_precs(self.block_precs(self.prec), self.prec)
        sp = tuple(wp[i] for i in conv_idx) * 2
        pb = self.__dict__.get('_train_packs')
        if pb is None or pb.prec != self.prec or pb.precs != sp or pb.key != tuple((w.data_ptr(), m, bool(k_)) for w, m, k_ in specs):
            pb = ops.PackBatch(specs, self.prec, sp)
            self.__dict__['_train_packs'] = pb
        allp = pb.update()
        packs, packsT = [None] * len(weights), [None] * len(weights)
        for j, i in enumerate(conv_idx):
            packs[i], packsT[i] = allp[j], allp[len(conv_idx) + j]
        return packs, packsT

    def _phase_weight_indices(self):
        """indices (into ``_conv_weights``' list) of the conv1 weights whose up block runs the phase-decomposed forms (``phase_conv``)"""
        cached = self.__dict__.get('_phase_idx')
        if cached is None:
            cached, wi, size = set(), 0, self.constant.constant.shape[-1]
            for cin, cout, up in self.blocks_cfg:
                size *= 2 if up else 1
                if phase_conv(up, size, self.reflect):
                    cached.add(wi)
                wi += 4 if (cin != cout or up) else 2
            self.__dict__['_phase_idx'] = cached
        return cached

    def prepare_step(self):
        """The parts of a TRAINING forward that depend on the weights only -- the spectral-norm power iteration and the 16-bit weight packs
        -- may be issued ahead of ``forward`` (on a side stream, beside the encoders: runners/holycow.py, streams.py).  ``forward`` picks
        the results up; without this call it computes them itself.  Same arithmetic either way."""
        if not (self.training and torch.is_grad_enabled() and next(self.parameters()).is_cuda):
            return
        states = self._sn_update()
        self.__dict__['_prepared_sn'] = states
        weights, sn = self._conv_weights(states)
        if any(w.requires_grad for w in weights):
            self.__dict__['_prepared_packs'] = self._train_packs_update(weights, sn)

    def _sn_layers(self):
        """batched SNWeight layers in a fixed order: decoder convs (block order, w1, w2[, skip]), head conv, projector.0, projector.2"""
        cached = self.__dict__.get('_sn_layer_cache')
        if cached is None:
            layers = []
            nb = len(self.blocks_cfg)
            for i in range(nb):
                c1, c2, sk = self.decoder_blocks._modules[str(i)].convs()
                layers += [c1, c2] + ([sk] if sk is not None else [])
            layers.append(self.decoder_blocks._modules[str(nb + 2)])
            layers += self._projector_sn_layers()
            cached = (layers, None)
            self.__dict__['_sn_layer_cache'] = cached
        return cached

    def forward(self, data_dict):
        affine = self._affine_params(data_dict)
        states = self._sn_states
        weights, sn = self._conv_weights(states)
        need_grad = torch.is_grad_enabled() and (affine.requires_grad or any(w.requires_grad for w in weights))
        prec = self.prec if need_grad else self.infer_prec
        precs = self.block_precs(prec)
        packs = packsT = None
        prepared = self.__dict__.pop('_prepared_packs', None)
        if not torch.is_grad_enabled():
            prepared = None
        if need_grad and self.training:
            packs, packsT = prepared if prepared is not None else self._train_packs_update(weights, sn)
        if not need_grad and not self.training:
            # inference (drive.py): the weights do not change between frames -> pack them to 16 bit once.  The key also carries the
            # generation counter of the fused optimizer / EMA kernels: those update weights through raw pointers without bumping
            # ``_version`` (a graph replay bumps neither: GraphedTrainStep.__call__ advances the counter itself).
            from .optim import WEIGHTS_GENERATION
            key = (prec, tuple(precs), WEIGHTS_GENERATION[0]) + tuple((w.data_ptr(), w._version) for w in weights)
            cache = self.__dict__.get('_pack_cache')
            if cache is None or cache[0] != key:
                ph = self._phase_weight_indices()
                wp = self._weight_precs(precs, prec)
                cache = (key, [ops.pack_weights(w.detach().contiguous(), 2 if i in ph else 0, wp[i]) if s_ is not None else None
                               for i, (w, s_) in enumerate(zip(weights, sn))])
                self.__dict__['_pack_cache'] = cache
            packs = cache[1]
        cfg = dict(blocks=self.blocks_cfg, prec=prec, precs=precs, need_grad=need_grad, sn=sn, packs=packs, packsT=packsT,
                   debug=getattr(self, '_debug', None), y16=G_Y16 and self.training and need_grad, reflect=self.reflect)
        rgbs, segm = _DecoderFunction.apply(cfg, affine, self.constant.constant, *weights)
        data_dict['fake_rgbs'] = rgbs
        data_dict['fake_segm'] = segm


class GeneratorFSTHPlus(Generator):
    """Drop-in for generators/FSTH_plus.py::Generator (BASELINE configs[4], 512 x 512): the same AdaIN decoder; the projector is three
    plain ``nn.Linear`` with LeakyReLU(0.05) in between (FSTH_plus.py:96-103, no spectral norm) and the pose vector is
    ``dec_keypoints[:, 0] - 0.5`` (FSTH_plus.py:129-139).  state_dict keys: ``affine_params_projector.{0,2,4}.{weight,bias}``."""

    def _build_projector(self, joint):
        hidden = max(512, joint)
        return _Indexed(_0=nn.Linear(joint, hidden), _2=nn.Linear(hidden, hidden), _4=nn.Linear(hidden, self.num_affine_params))

    def _projector_sn_layers(self):
        return []

    def _pose_vector(self, data_dict):
        return data_dict['dec_keypoints'][:, 0] - 0.5

    def _project(self, joint, states):
        """three plain Linear layers on the lp_linear_* weight-stream kernels (no library GEMM); LeakyReLU(0.05) on B x 648 values stays a torch op"""
        m = self.affine_params_projector._modules
        h = F.leaky_relu(hip_linear(joint, m['0'].weight, m['0'].bias), 0.05)
        h = F.leaky_relu(hip_linear(h, m['2'].weight, m['2'].bias), 0.05)
        return hip_linear(h, m['4'].weight, m['4'].bias)


# ----------------------------------------------------------------------------------------------------------------------
# generic HIP layers for the discriminator and the VGG criterions (NHWC fp32 tensors)
# ----------------------------------------------------------------------------------------------------------------------
def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """logical NCHW -> contiguous [N,H,W,C] (zero-copy when the tensor already has channels_last strides)"""
    return x.permute(0, 2, 3, 1).contiguous()


def as_nchw_view(x_nhwc: torch.Tensor) -> torch.Tensor:
    """[N,H,W,C] storage presented with the reference's logical N x C x H x W shape (channels_last strides, no copy)"""
    return x_nhwc.permute(0, 3, 1, 2)


class ConvFn(torch.autograd.Function):
    """y = conv_{k x k, pad k//2}(act(x), w) + bias + res,  act = identity (pro=0) | ReLU (pro=2);  x, res, y NHWC fp32.
    Forward = lp_act_pack (act(x) -> 16-bit operand planes, kept for backward) + lp_conv16_fwd; backward = one lp_act_pack of dY
    feeding both lp_conv16_fwd with the flipped/transposed pack (data gradient; the ReLU mask is read from the saved planes in its
    epilogue) and lp_conv16_wgrad (which also emits the bias gradient).  Convs with <= 4 channels on one side use the fp32
    thin-channel kernels.  ``packs`` = optional cached (forward, dgrad) WeightPacks of a frozen w."""

    @staticmethod
    def forward(ctx, x, w, bias, res, ksize, pro, prec, packs, sn=None, x16=None, emit=None, reflect=False):
        """``sn`` = (u_used, v_used, [sigma, 1/sigma]) from SNBatch when ``w`` is a spectrally normalised W_orig.
        ``x16``: operand planes of act(x) that already exist (emitted by the producer conv's epilogue, or packed once for several
        consumers) -- skips this call's lp_act_pack.  ``emit`` = (out_relu: 0|1, holder list): the conv epilogue also writes the
        operand planes of (relu?)(y) for the consumer conv; they are appended to ``holder``.  ``reflect``: the 3x3 conv sits behind
        nn.ReflectionPad2d(1) instead of zero padding (blocks.py:76-88, --dis_padding reflection): the zero-padded MFMA conv + the border terms of
        csrc/reflect_border.hip, forward and backward; no emitted planes (they would be taken before the correction)."""
        assert not reflect or (ksize == 3 and emit is None), 'reflection padding: 3x3 convs without epilogue-emitted planes'
        small_k = ksize == 3 and w.shape[1] <= 32
        wd = w.detach().contiguous()
        # planes-only output (emit = (relu, holder, emit_prec | None, False)): nothing reads the fp32 y of a conv whose only consumer takes the
        # emitted planes (conv -> ReLU -> conv chains of the critic) -- the autograd edge is carried by a zero-stride phantom of y's shape
        want_y = not (emit is not None and len(emit) > 3 and emit[3] is False)
        if isinstance(packs, dict):          # per-step cache shared by several calls on the same W_orig (discriminator passes)
            key = (wd.data_ptr(), 0)
            if key not in packs:
                packs[key] = ops.pack_weights(wd, 0, prec, small_k=small_k)
            pack = packs[key]
        else:
            pack = packs[0] if packs is not None else ops.pack_weights(wd, 0, prec, small_k=small_k)
        cin, cout, width = x.shape[3], wd.shape[0], x.shape[2]
        bd = None if bias is None else bias.detach().contiguous()
        alpha = None if sn is None else sn[2][1:]
        need_w = w.requires_grad
        thin_w = need_w and not reflect and ops.thin_wgrad_supported(cin, cout, ksize, pro, width)       # the weight gradient will want fp32 x
        a16 = x16
        if a16 is None and pro == 0 and res is None and not reflect and ops.thin_conv_supported(cin, cout, ksize, width):
            y = ops.thin_conv(x, pack, ksize=ksize, bias=bd, alpha=alpha, prec=prec, out16=None if emit is None else emit[0],
                              out16_prec=None if emit is None or len(emit) < 3 else emit[2], want_y=want_y)
            if emit is not None:
                y, o16 = y
                emit[1].append(o16)
                if emit[0]:
                    tape_relu(lambda: emit[1][-1].hi[..., :cout] > 0, True)
        else:
            if a16 is None:
                a16 = ops.act_pack(x, pro=pro, prec=prec)
                if pro == 2:
                    tape_relu(lambda: a16.hi[..., :cin] > 0, True)
            #  the MFMA path emits the consumer planes in its OWN operand mode; another mode is only served by the thin-channel branch
            assert emit is None or len(emit) < 3 or emit[2] in (None, prec), 'emit_prec differs from the conv\'s operand mode on the MFMA path'
            y = ops.conv16(a16, pack, ksize=ksize, bias=bd, res=res, alpha=alpha, prec=prec, out16=None if emit is None else emit[0],
                           want_y=want_y or cout % 8 != 0)
            if reflect:
                ops.reflect_border_fwd(a16, wd, alpha, y, prec=prec)
            if emit is not None:
                y, o16 = y
                emit[1].append(o16)
                if emit[0]:
                    tape_relu(lambda: o16.hi[..., :cout] > 0, True)
        if y is None:
            y = phantom((x.shape[0], x.shape[1], x.shape[2], cout), x.device)
        if need_w and not thin_w and a16 is None:
            a16 = ops.act_pack(x, pro=pro, prec=prec)
        ctx.reflect = reflect
        ctx.x = x if thin_w else None                               # fp32 input only where a thin-channel weight gradient needs it
        ctx.a16 = a16 if ((need_w and not thin_w) or pro == 2) else None
        ctx.wd = wd
        ctx.w_param = w if (sn is not None and w.requires_grad and w.is_leaf) else None
        ctx.b_param = bias if (bias is not None and bias.requires_grad and bias.is_leaf) else None
        ctx.accum_alt = _ALT['on']
        ctx.cfg = (ksize, pro, prec, packs, bias is not None, res is not None, sn)
        return y

    @staticmethod
    def backward(ctx, dy):
        wd, a16, x = ctx.wd, ctx.a16, ctx.x
        ksize, pro, prec, packs, has_bias, has_res, sn = ctx.cfg
        dy = dy.contiguous()
        dx = dw = db = dres = None
        d16 = None

        def dy16():
            nonlocal d16
            if d16 is None:
                rec = getattr(dy, '_lp_planes', None)          # the producer of dy already wrote its operand planes (nn.ConvPoolFn.backward -> the skip conv)
                if rec is not None and rec[1] == prec and rec[2] == dy._version:
                    d16 = rec[0]
                else:
                    d16 = ops.act_pack(dy, prec=prec, grad=True)
            return d16
        cout, cin, width = dy.shape[3], wd.shape[1], dy.shape[2]
        if ctx.needs_input_grad[0]:
            if isinstance(packs, dict):
                key = (wd.data_ptr(), 1)
                if key not in packs:
                    packs[key] = ops.pack_weights(wd, 1, prec, small_k=(ksize == 3 and wd.shape[0] <= 32))
                packT = packs[key]
            else:
                packT = packs[1] if packs is not None else ops.pack_weights(wd, 1, prec, small_k=(ksize == 3 and wd.shape[0] <= 32))
            alpha = None if sn is None else sn[2][1:]
            if pro != 2 and ops.thin_conv_supported(cout, cin, ksize, width):
                dx = ops.thin_conv(dy, packT, ksize=ksize, alpha=alpha, prec=prec)
            else:
                # pro == 2: the forward applied ReLU to x first -> dx = dA * (x > 0), fused into the dgrad launch's epilogue
                dx = ops.conv16(dy16(), packT, ksize=ksize, alpha=alpha, prec=prec, relu_mask=a16 if pro == 2 else None, amax=not ctx.reflect)
            if ctx.reflect:          # (the border terms change dx after the launch: its folded max|dx| would be stale -> the consumer takes its own)
                ops.reflect_border_dgrad(dy, wd, alpha, dx, a16 if pro == 2 else None)
        want_db = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            kw = dict(ksize=ksize, sn=None if sn is None else (wd,) + tuple(sn),
                      accum=None if ctx.w_param is None else _accum_target(ctx.w_param, ctx.accum_alt), bias_grad=want_db)
            if x is not None:
                dw = ops.thin_wgrad(x, dy, pro=pro, **kw)
            else:      # (fused accumulation: the reduce launch adds the bias gradient straight into the bias parameter's .grad)
                dw = ops.conv_wgrad16(a16, dy16(), prec=prec, bias_accum=_accum_target(ctx.b_param, ctx.accum_alt) if (want_db and ctx.b_param is not None) else None,
                                      **kw)
            if want_db:
                dw, db = dw
            if ctx.reflect:
                corr = ops.reflect_border_wgrad(a16, dy, prec=prec, sn=kw['sn'], accum=kw['accum'])
                if corr is not None:
                    dw = dw + corr
        elif want_db:
            db = dy.sum(dim=(0, 1, 2))
        if has_res and ctx.needs_input_grad[3]:
            dres = dy
        return dx, dw, db, dres, None, None, None, None, None, None, None, None


def hip_conv(x, w, bias=None, res=None, ksize=3, pro=0, prec=None, packs=None, sn=None, x16=None, emit16=None, emit_prec=None, want_y=True, reflect=False):
    """``x16``: existing operand planes of act(x); ``emit16`` = 0 | 1: also return the operand planes of y (1: of relu(y)), written
    by the conv's epilogue -> ``(y, Act16)``; ``emit_prec``: their operand mode when the consumer's differs from this conv's ``prec``
    (thin-channel first layers only).  ``want_y=False`` (with ``emit16``): the returned y is a PHANTOM (shape and autograd edge only, no
    storage) -- for convs whose fp32 output nobody reads.  ``reflect``: reflection instead of zero padding (3x3, no ``emit16``)."""
    prec = default_prec() if prec is None else prec
    _TAPE_GRAD[0] = torch.is_grad_enabled()
    if emit16 is None:
        return ConvFn.apply(x, w, bias, res, ksize, pro, prec, packs, sn, x16, None, reflect)
    holder = []
    y = ConvFn.apply(x, w, bias, res, ksize, pro, prec, packs, sn, x16, (int(emit16), holder) + ((() if emit_prec is None else (emit_prec,)) if want_y else (emit_prec, False)))
    return y, holder[0]


class ConvPoolFn(torch.autograd.Function):
    """y = [relu] AvgPool2d(2)( conv3x3(relu(h), W) + bias ) + res  as ONE launch (round 6): conv followed by the 2 x 2 average is a 4x4 stride-2
    conv -- 16 taps per pooled output instead of 4 x 9 (4/9 of the matrix work), and the full-resolution conv output is never written or read.
    The critic's down blocks and stem (discriminators/no_landmarks.py:52-81 of the reference; blocks.py:76-90): out = conv2(relu(h)) + skip(x),
    AvgPool2d(2), then the next block's in-place ReLU -- the pooled skip branch comes in as ``res`` (a 1x1 conv commutes with the pool: it runs on
    the pooled block input, a quarter of its work).  ``x16`` = the operand planes of relu(h) (conv1's epilogue wrote them; ``h`` itself may be
    a phantom); ``relu_out``: the result is relu(pooled) -- what every consumer of a block output sees (blocks.py:71-73) -- and ``emit`` (a list)
    receives its operand planes for the next conv.  Forward: lp_conv16_fwd upsample = 3 with an lp_pack_weights mode-4 image.  Backward: the data
    gradient (with the ReLU mask of relu(h)) is the scatter-free phase-forward form (upsample = 2, mode-5 image) on the planes of the masked
    pooled gradient; the weight / bias gradient takes the 0.25-upsampled gradient through lp_conv16_wgrad as the unfused chain does."""

    @staticmethod
    def forward(ctx, h, w, bias, res, prec, packs, sn, x16, relu_out, emit):
        wd = w.detach().contiguous()

        def pk(mode):
            if isinstance(packs, dict):
                key = (wd.data_ptr(), mode)
                if key not in packs:
                    packs[key] = ops.pack_weights(wd, mode, prec)
                return packs[key]
            return ops.pack_weights(wd, mode, prec)
        bd = None if bias is None else bias.detach().contiguous()
        alpha = None if sn is None else sn[2][1:]
        out = ops.conv16(x16, pk(4), ksize=3, bias=bd, res=res, alpha=alpha, prec=prec, phase_dgrad=True,
                         out16=(1 if relu_out else 0) if emit is not None else None, y_relu=bool(relu_out))
        if emit is not None:
            y, o16 = out
            emit.append(o16)
        else:
            y = out
        ctx.x16, ctx.wd, ctx.pk = x16, wd, pk
        ctx.save_for_backward(y if relu_out else None)          # (an OUTPUT kept on ctx as a plain attribute would be a reference cycle: the graph -- and every
                                                                 #  spectral-norm state set it reserves -- would only be freed by the garbage collector)
        ctx.w_param = w if (sn is not None and w.requires_grad and w.is_leaf) else None
        ctx.b_param = bias if (bias is not None and bias.requires_grad and bias.is_leaf) else None
        ctx.accum_alt = _ALT['on']
        ctx.cfg = (prec, bias is not None, res is not None, sn)
        return y

    @staticmethod
    def backward(ctx, dy):
        prec, has_bias, has_res, sn = ctx.cfg
        x16, wd = ctx.x16, ctx.wd
        (y_out,) = ctx.saved_tensors
        dy = dy.contiguous()
        f16 = prec == PREC_F16
        dx = dw = db = None
        alpha = None if sn is None else sn[2][1:]
        want_db = has_bias and ctx.needs_input_grad[2]
        want_res = has_res and ctx.needs_input_grad[3]
        if ops.POOL_GRAD_FUSED and dy.shape[3] % 8 == 0:
            # ONE pass over the pooled gradient: the ReLU behind the pool, the planes of the masked gradient (data-gradient operand; the skip conv's
            # backward picks them up through ``_lp_planes``) and the planes of the pool's adjoint 0.25 * up2(dm) (weight-gradient operand) -- no
            # full-resolution fp32 gradient is written or read
            need_dm = want_res or (want_db and not ctx.needs_input_grad[1])
            dm, d16, up16 = ops.pool_grad_pack(dy, y_out, prec, want_dm=need_dm and y_out is not None, want_lo=ctx.needs_input_grad[0] or want_res,
                                               want_up=ctx.needs_input_grad[1])
            if y_out is None:
                dm = dy          # (no ReLU behind the pool: the masked gradient IS dy)
            if dm is not None and d16 is not None:
                dm._lp_planes = (d16, prec, dm._version)
        else:
            dm = dy if y_out is None else torch.where(y_out > 0, dy, torch.zeros((), dtype=dy.dtype, device=dy.device))      # the ReLU behind the pool
            d16 = ops.act_pack(dm, prec=prec, grad=True) if ctx.needs_input_grad[0] else None
            up16 = None
            if ctx.needs_input_grad[1]:
                dhi = ops.avgpool2_bwd(dy, None, False, amax=f16, y_relu=y_out)          # 0.25 * nearest-upsampled (masked) gradient: the conv output's
                up16 = ops.act_pack(dhi, prec=prec, grad=True)
        if ctx.needs_input_grad[0]:
            dx = ops.conv16(d16, ctx.pk(5), ksize=3, upsample=True, phase=True, alpha=alpha, prec=prec, relu_mask=x16, amax=f16)
        if ctx.needs_input_grad[1]:
            dw = ops.conv_wgrad16(x16, up16, ksize=3, prec=prec, sn=None if sn is None else (wd,) + tuple(sn),
                                  accum=None if ctx.w_param is None else _accum_target(ctx.w_param, ctx.accum_alt), bias_grad=want_db,
                                  bias_accum=_accum_target(ctx.b_param, ctx.accum_alt) if (want_db and ctx.b_param is not None) else None)
            if want_db:
                dw, db = dw
        elif want_db:
            db = dm.sum(dim=(0, 1, 2))
        return dx, dw, db, (dm if want_res else None), None, None, None, None, None, None


class AvgPool2Fn(torch.autograd.Function):
    """AvgPool2d(2) of relu?(x), NHWC (blocks.py:89-90 / perceptual_loss.py:77 with the preceding ReLU fused); ``relu_out``: relu(pool(x)) -- the
    in-place ReLU of the critic's NEXT block (blocks.py:71-73) fused into the pool launch (round 5: one launch instead of pool + relu + pack)."""

    @staticmethod
    def forward(ctx, x, relu_in, emit=None, relu_out=False):
        """``emit`` = (prec, holder list): the pool launch also writes the operand planes of y for the conv that follows"""
        ctx.relu_in = relu_in
        ctx.f16 = (default_prec() if emit is None else emit[0]) == PREC_F16      # the mode of the module that owns this pool
        if emit is None:
            y = ops.avgpool2_fwd(x, relu_in, relu_out=relu_out)
        else:
            y, o16 = ops.avgpool2_fwd(x, relu_in, out16_prec=emit[0], relu_out=relu_out)
            emit[1].append(o16)
        # only what the backward reads is kept: the input for a ReLU on the way in, the OUTPUT for a ReLU on the way out, else nothing
        ctx.kept = 'x' if relu_in else 'y' if relu_out else None
        ctx.save_for_backward(x if relu_in else y if relu_out else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (t,) = ctx.saved_tensors
        return ops.avgpool2_bwd(dy.contiguous(), t if ctx.kept == 'x' else None, ctx.relu_in, amax=ctx.f16, y_relu=t if ctx.kept == 'y' else None), None, None, None


class L1Fn(torch.autograd.Function):
    """mean |relu?(a) - relu?(b)|  (F.l1_loss with the preceding ReLUs fused); gradient only w.r.t. ``a`` (b is detached in
    every call site of the reference: featmat.py:17, perceptual_loss.py:93)."""

    @staticmethod
    def forward(ctx, a, b, relu_in):
        if not ctx.needs_input_grad[0]:
            return ops.l1_sum(a, b, relu_in, 1.0 / a.numel())
        term, ctx.sgn = ops.l1_sum(a, b, relu_in, 1.0 / a.numel(), want_sign=True)      # the backward reads the 1-byte sign pattern,
        ctx.shape = tuple(a.shape)                                                        # not a and b again
        ctx.f16 = default_prec() == PREC_F16
        return term

    @staticmethod
    def backward(ctx, g):
        return ops.l1_bwd(None, None, g, 1.0 / ctx.sgn.numel(), False, sign=ctx.sgn, shape=ctx.shape, amax=ctx.f16), None, None


def hip_l1(a, b, relu_in=False):
    return L1Fn.apply(a, b.detach(), relu_in)


def hip_l1_mean(a, b):
    """``F.l1_loss(a, b.detach())`` (mean reduction) of two equally shaped fp32 CUDA tensors of ANY shape on the lp_l1_* kernels: the tensors are
    taken in their storage order (an element-wise loss does not care), flattened, and -- toy shapes only -- zero-padded to a multiple of 4"""
    if not (a.is_cuda and b.is_cuda):
        raise RuntimeError('the L1 criterions run on the MI355X HIP path only (no CPU fallback)')
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.dim() == 4 and a.stride() == b.stride() and a.permute(0, 2, 3, 1).is_contiguous():      # channels_last views (the critic's feature lists): no copy
        a, b = a.permute(0, 2, 3, 1), b.permute(0, 2, 3, 1)
    a, b = a.contiguous().reshape(-1), b.detach().contiguous().reshape(-1)
    n = a.numel()
    if n % 4:
        a, b = F.pad(a, (0, 4 - n % 4)), F.pad(b, (0, 4 - n % 4))
        return L1Fn.apply(a, b, False) * (a.numel() / n)
    return L1Fn.apply(a, b, False)


class L1TapFn(torch.autograd.Function):
    """A feature tap of the perceptual stacks (perceptual_loss.py:86-108): returns ``(a, mean|relu?(a) - relu?(b)|)`` where the
    first output is ``a`` itself, to be handed to the next layer.  ``a`` then has two consumers -- the L1 term and the next
    conv / pool -- and autograd would sum their two gradients with a separate ``add`` pass over the feature map; here the sum
    happens inside the L1 backward kernel (its optional ``add`` input)."""

    @staticmethod
    def forward(ctx, a, b, relu_in):
        if not ctx.needs_input_grad[0]:
            return a.view_as(a), ops.l1_sum(a, b, relu_in, 1.0 / a.numel())
        term, ctx.sgn = ops.l1_sum(a, b, relu_in, 1.0 / a.numel(), want_sign=True)
        ctx.shape = tuple(a.shape)
        ctx.f16 = default_prec() == PREC_F16
        return a.view_as(a), term

    @staticmethod
    def backward(ctx, g_next, g_loss):
        if g_loss is None:
            return g_next, None, None
        add = None if g_next is None else g_next.contiguous()
        return ops.l1_bwd(None, None, g_loss, 1.0 / ctx.sgn.numel(), False, add=add, sign=ctx.sgn, shape=ctx.shape,
                          amax=ctx.f16), None, None


def phantom(shape, device):
    """a tensor that carries a shape and an autograd edge but no storage (stride 0): planes-only chains hand it from Function to Function"""
    return torch.empty(1, dtype=torch.float32, device=device).expand(*shape)          # (never read: no fill launch)


class L1Tap16Fn(torch.autograd.Function):
    """``L1TapFn`` for a planes-only chain: the tap is the 16-bit operand planes of relu(y) (``a16``), ``a`` is the phantom that carries the
    autograd edge of y.  Backward as L1TapFn (the sign pattern of the forward carries the ReLU mask)."""

    @staticmethod
    def forward(ctx, a, a16, b16):
        if not ctx.needs_input_grad[0]:
            return a.view_as(a), ops.l1_sum16(a16, b16, 1.0 / a16.act.hi.numel())
        term, ctx.sgn = ops.l1_sum16(a16, b16, 1.0 / a16.act.hi.numel(), want_sign=True)
        ctx.shape = tuple(a16.act.hi.shape)
        ctx.f16 = a16.prec == PREC_F16
        return a.view_as(a), term

    @staticmethod
    def backward(ctx, g_next, g_loss):
        if g_loss is None:
            return g_next, None, None
        add = None if g_next is None else g_next.contiguous()
        return ops.l1_bwd(None, None, g_loss, 1.0 / ctx.sgn.numel(), False, add=add, sign=ctx.sgn, shape=ctx.shape, amax=ctx.f16), None, None


class AvgPool2Fn16(torch.autograd.Function):
    """AvgPool2d(2) of relu(x) in a planes-only chain: x16 = operand planes of relu(x) -> the planes of the pooled tensor (appended to
    ``holder``); x and the result are phantoms.  Backward: 0.25 * dy * [x16 > 0]."""

    @staticmethod
    def forward(ctx, x, x16, prec, holder):
        o16 = ops.avgpool2_fwd16(x16, prec)
        holder.append(o16)
        ctx.x16 = x16
        ctx.f16 = prec == PREC_F16
        return phantom(o16.hi.shape, o16.hi.device)

    @staticmethod
    def backward(ctx, dy):
        return ops.avgpool2_bwd_m16(dy.contiguous(), ctx.x16, amax=ctx.f16), None, None, None


def hip_l1_tap16(a, a16, b16):
    """planes-only form of ``hip_l1_tap``: -> (phantom passed through, l1 term)"""
    return L1Tap16Fn.apply(a, a16, b16)


def hip_l1_tap(a, b, relu_in=False):
    """-> (a passed through, l1 term); use the returned tensor as the input of the following layer.  ``b``: fp32 tensor, or ``ops.Tap16``
    (the other image's tap kept as the 16-bit operand planes of relu(y))"""
    if RELU_TAPE is not None:       # sign pattern of this L1 site (tie-masked parity tests)
        bb = b.float() if isinstance(b, ops.Tap16) else b
        tape_relu(lambda: torch.sign((torch.relu(a) if relu_in else a) - (torch.relu(bb) if relu_in else bb)).to(torch.int8))
    return L1TapFn.apply(a, b.detach(), relu_in)
