"""Tensor-level wrappers around the C ABI.  Activations are NHWC fp32 contiguous CUDA tensors of shape [N,H,W,C].
PyTorch here is plumbing only: it owns the buffers and the stream; all arithmetic happens in liblp_hip.so."""
from typing import NamedTuple, Optional, Tuple

import torch

from . import _lib
from ._lib import PREC_BF16, PREC_BF16X3, check

Tensor = torch.Tensor


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Optional live profiling hook used by bench.py: when PROFILE is a list, every conv / wgrad launch is bracketed by
# HIP events recorded on the launch stream and appended as (kind, flops, start_event, end_event).
PROFILE = None


class _Timed:
    def __init__(self, kind, flops, tag=None):
        self.kind, self.flops, self.tag = kind, flops, tag

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.e1.record()
            PROFILE.append((self.kind, self.flops, self.e0, self.e1, self.tag))


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: Tensor, name: str):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f'{name}: expected a contiguous fp32 CUDA tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}')


def _round_up(v, m):
    return (v + m - 1) // m * m


class WeightPack(NamedTuple):
    hi: Tensor
    lo: Optional[Tensor]
    rows: int      # logical rows (Cout for mode 0, Cin for mode 1)
    cols: int
    rows_p: int
    cols_p: int
    taps: int


def pack_weights(w: Tensor, mode: int, prec: int, small_k: bool = False) -> WeightPack:
    """w: [Cout, Cin, k, k] (or [Cout, Cin]) fp32.  mode 0 = forward pack, mode 1 = dgrad pack (flipped + transposed).
    small_k pads the contraction dim to 32 instead of 64 (only valid for 3x3 non-upsampled convs)."""
    _chk(w, 'w')
    cout, cin = w.shape[0], w.shape[1]
    taps = w.numel() // (cout * cin)
    rows, cols = (cout, cin) if mode == 0 else (cin, cout)
    rows_p = _round_up(rows, 128)
    cols_p = _round_up(cols, 32 if (small_k and cols <= 32) else 64)
    hi = torch.empty((taps, rows_p, cols_p), dtype=torch.int16, device=w.device)
    lo = torch.empty_like(hi) if prec == PREC_BF16X3 else None
    check(_lib.lib().lp_pack_weights(w.data_ptr(), hi.data_ptr(), _p(lo), cout, cin, taps, rows_p, cols_p, mode, _stream()),
          'lp_pack_weights')
    return WeightPack(hi, lo, rows, cols, rows_p, cols_p, taps)


class PackBatch:
    """bf16 packs of MANY conv weights, refreshed by ONE launch (lp_pack_weights_batch).  ``specs`` = [(w, mode, small_k)];
    the pack buffers and the device descriptor table are allocated once (static addresses: hipGraph friendly); ``update()``
    re-packs all entries from the current weight values and returns the list of WeightPacks (same order as ``specs``)."""

    def __init__(self, specs, prec: int):
        import struct
        assert _lib.lib().lp_pack_desc_bytes() == 56
        self.prec = prec
        self.key = tuple((w.data_ptr(), mode, bool(sk)) for w, mode, sk in specs)
        self.packs = []
        blob = bytearray()
        self.chunks = 0
        for w, mode, small_k in specs:
            _chk(w, 'w')
            cout, cin = w.shape[0], w.shape[1]
            taps = w.numel() // (cout * cin)
            rows, cols = (cout, cin) if mode == 0 else (cin, cout)
            rows_p = _round_up(rows, 128)
            cols_p = _round_up(cols, 32 if (small_k and cols <= 32) else 64)
            hi = torch.empty((taps, rows_p, cols_p), dtype=torch.int16, device=w.device)
            lo = torch.empty_like(hi) if prec == PREC_BF16X3 else None
            self.packs.append(WeightPack(hi, lo, rows, cols, rows_p, cols_p, taps))
            blob += struct.pack('<QQQiiiiiiii', w.data_ptr(), hi.data_ptr(), 0 if lo is None else lo.data_ptr(), cout, cin, taps, rows_p,
                                cols_p, mode, self.chunks, 0)
            self.chunks += (taps * rows_p * cols_p + 1023) // 1024
        self.table = torch.frombuffer(blob, dtype=torch.uint8).clone().to(specs[0][0].device)

    def update(self):
        check(_lib.lib().lp_pack_weights_batch(self.table.data_ptr(), len(self.packs), self.chunks, _stream()), 'lp_pack_weights_batch')
        return self.packs


def conv(x: Tensor, pack: WeightPack, *, ksize: int, upsample: bool = False, pro: int = 0, scale: Optional[Tensor] = None,
         shift: Optional[Tensor] = None, bias: Optional[Tensor] = None, res: Optional[Tensor] = None, res_shift: int = 0,
         alpha: Optional[Tensor] = None, prec: int = PREC_BF16, relu_mask: Optional[Tensor] = None) -> Tensor:
    """y = alpha * conv(up2?(act(x)), pack) + bias + res ; x [N,Hin,Win,Cin] -> y [N,H,W,Cout].
    ``relu_mask`` [N,H,W,Cout]: y is zeroed where relu_mask <= 0 (fused ReLU backward when this launch is a data gradient)."""
    _chk(x, 'x')
    n, hin, win, cin = x.shape
    assert cin == pack.cols and pack.taps == ksize * ksize, (x.shape, pack.rows, pack.cols, pack.taps)
    h, w = (hin * 2, win * 2) if upsample else (hin, win)
    cout = pack.rows
    y = torch.empty((n, h, w, cout), dtype=torch.float32, device=x.device)
    if relu_mask is not None:
        assert relu_mask.shape == (n, h, w, cout), (relu_mask.shape, (n, h, w, cout))
    for t, nm in ((scale, 'scale'), (shift, 'shift'), (bias, 'bias'), (res, 'res'), (relu_mask, 'relu_mask')):
        if t is not None:
            _chk(t, nm)
    if res is not None:
        assert res.shape == (n, h >> res_shift, w >> res_shift, cout), (res.shape, y.shape, res_shift)
    with _Timed('conv_igemm', 2.0 * n * h * w * cout * cin * ksize * ksize, (n, h, w, cin, cout, ksize, int(upsample), pro)):
        check(_lib.lib().lp_conv_fwd(x.data_ptr(), pack.hi.data_ptr(), _p(pack.lo), y.data_ptr(), _p(scale), _p(shift), _p(bias),
                                     _p(res), _p(alpha), n, h, w, cin, cout, pack.cols_p, pack.rows_p, ksize, int(upsample), pro,
                                     res_shift, prec, _p(relu_mask), _stream()), 'lp_conv_fwd')
    return y


def conv_wgrad(x: Tensor, dy: Tensor, *, ksize: int, upsample: bool = False, pro: int = 0, scale: Optional[Tensor] = None,
               shift: Optional[Tensor] = None, prec: int = PREC_BF16, splits: Optional[int] = None, sn=None,
               accum: Optional[Tensor] = None, bias_grad: bool = False):
    """dw [Cout,Cin,k,k] = sum_pixels dy (x) up2?(act(x)) (shifted by tap).
    ``sn`` = (w_orig, u, v, sig): the layer is spectrally normalised (forward used alpha = 1/sigma in the conv epilogue); the
    returned gradient is then w.r.t. W_orig: dw/sigma - <dw, W_orig>/sigma^2 u v^T.
    ``accum`` (with ``sn``): add that gradient to this tensor (the parameter's .grad) instead and return None.
    ``bias_grad``: return ``(dw, db)`` with db [Cout] = sum_pixels dy, produced by the same launch (the kernel streams dy anyway)
    or, for the <= 4-channel head where the kernel does not emit it, by a column sum."""
    _chk(x, 'x'); _chk(dy, 'dy')
    n, h, w, cout = dy.shape
    cin = x.shape[3]
    if splits is None:
        blocks = _round_up(cout, 64) // 64 * (_round_up(cin, 64) // 64)
        splits = max(1, min(512 // blocks, (n * h * w + 127) // 128))     # ~2 workgroups per CU in total
    ws_bytes = _lib.lib().lp_conv_wgrad_workspace_bytes(cin, cout, ksize, splits)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
    dw = torch.empty((cout, cin, ksize, ksize), dtype=torch.float32, device=x.device)
    dot = torch.empty(512, dtype=torch.float32, device=x.device) if sn is not None else None      # per-block partials of <g, W>
    db = None
    if bias_grad and _lib.lib().lp_conv_wgrad_has_dbias(cin, cout, ksize, int(upsample), pro):
        db = torch.empty(cout, dtype=torch.float32, device=x.device)
    with _Timed('conv_wgrad', 2.0 * n * h * w * cout * cin * ksize * ksize, (n, h, w, cin, cout, ksize, int(upsample), pro)):
        check(_lib.lib().lp_conv_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), _p(scale), _p(shift), n, h, w, cin,
                                       cout, ksize, int(upsample), pro, splits, prec, _p(db), _stream()), 'lp_conv_wgrad')
    if bias_grad and db is None:
        db = dy.sum(dim=(0, 1, 2))
    if sn is not None:
        w_orig, u, v, sig = sn
        if accum is not None:
            assert accum.is_contiguous() and accum.dtype == torch.float32 and accum.numel() == dw.numel()
        check(_lib.lib().lp_sn_grad_apply(dw.data_ptr(), w_orig.data_ptr(), u.data_ptr(), v.data_ptr(), sig.data_ptr(), dot.data_ptr(),
                                          _p(accum), cout, cin * ksize * ksize, _stream()), 'lp_sn_grad_apply')
        if accum is not None:
            dw = None
    return (dw, db) if bias_grad else dw


def sn_grad_apply(g: Tensor, w_orig: Tensor, u: Tensor, v: Tensor, sig: Tensor, accum: Optional[Tensor] = None) -> Optional[Tensor]:
    """gradient w.r.t. W_orig of a spectrally normalised layer from the raw gradient ``g`` w.r.t. W/sigma (legacy-hook autograd,
    u and v constants): g/sigma - <g, W_orig>/sigma^2 u v^T, in place on ``g`` -- or added to ``accum`` (returns None)."""
    _chk(g, 'g'); _chk(w_orig, 'w_orig')
    rows = g.shape[0]
    cols = g.numel() // rows
    dot = torch.empty(512, dtype=torch.float32, device=g.device)
    if accum is not None:
        assert accum.is_contiguous() and accum.dtype == torch.float32 and accum.numel() == g.numel()
    check(_lib.lib().lp_sn_grad_apply(g.data_ptr(), w_orig.data_ptr(), u.data_ptr(), v.data_ptr(), sig.data_ptr(), dot.data_ptr(),
                                      _p(accum), rows, cols, _stream()), 'lp_sn_grad_apply')
    return None if accum is not None else g


def instnorm_stats(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], eps: float
                   ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """x [N,H,W,C]; gamma/beta: [N,C] views (last dim contiguous) of the projector output.  -> mean, rstd, scale, shift [N,C]."""
    _chk(x, 'x')
    n, h, w, c = x.shape
    ab_stride = 0
    if gamma is not None:
        assert gamma.shape == (n, c) and beta.shape == (n, c) and gamma.stride(1) == 1 and beta.stride(1) == 1
        assert gamma.stride(0) == beta.stride(0)
        ab_stride = gamma.stride(0)
    mean, rstd, scale, shift = (torch.empty((n, c), dtype=torch.float32, device=x.device) for _ in range(4))
    ws = torch.empty(_lib.lib().lp_instnorm_workspace_bytes(n, h * w, c) // 4, dtype=torch.float32, device=x.device)
    check(_lib.lib().lp_instnorm_stats(x.data_ptr(), _p(gamma), _p(beta), ab_stride, eps, mean.data_ptr(), rstd.data_ptr(),
                                       scale.data_ptr(), shift.data_ptr(), ws.data_ptr(), n, h * w, c, _stream()), 'lp_instnorm_stats')
    return mean, rstd, scale, shift


def adain_relu_bwd(dA: Tensor, x: Tensor, add: Optional[Tensor], gamma: Tensor, mean: Tensor, rstd: Tensor, scale: Tensor,
                   shift: Tensor, dgamma: Tensor, dbeta: Tensor, upsample: bool) -> Tensor:
    """Backward of relu(AdaIN(x)) (+x2 upsample).  dgamma/dbeta: [N,C] views into the projector-output gradient (written)."""
    _chk(dA, 'dA'); _chk(x, 'x')
    n, h, w, c = x.shape
    assert dA.shape == (n, h << int(upsample), w << int(upsample), c), (dA.shape, x.shape)
    assert gamma.stride(1) == 1 and dgamma.stride(1) == 1 and dbeta.stride(1) == 1
    assert gamma.stride(0) == dgamma.stride(0) == dbeta.stride(0)
    dx = torch.empty_like(x)
    ws = torch.empty(_lib.lib().lp_adain_bwd_workspace_bytes(n, h * w, c) // 4, dtype=torch.float32, device=x.device)
    check(_lib.lib().lp_adain_relu_bwd(dA.data_ptr(), x.data_ptr(), _p(add), gamma.data_ptr(), gamma.stride(0), mean.data_ptr(),
                                       rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), dx.data_ptr(), dgamma.data_ptr(),
                                       dbeta.data_ptr(), ws.data_ptr(), n, h, w, c, int(upsample), _stream()), 'lp_adain_relu_bwd')
    return dx


def sum2x2(x: Tensor) -> Tensor:
    _chk(x, 'x')
    n, h2, w2, c = x.shape
    out = torch.empty((n, h2 // 2, w2 // 2, c), dtype=torch.float32, device=x.device)
    check(_lib.lib().lp_sum2x2(x.data_ptr(), out.data_ptr(), n, h2 // 2, w2 // 2, c, _stream()), 'lp_sum2x2')
    return out


def head_fwd(z: Tensor, want_t: bool = True) -> Tuple[Optional[Tensor], Tensor, Tensor]:
    """z [N,H,W,4] -> (tanh(z) NHWC, fake_rgbs NCHW [N,3,H,W], fake_segm NCHW [N,1,H,W])."""
    _chk(z, 'z')
    n, h, w, c = z.shape
    assert c == 4
    t = torch.empty_like(z) if want_t else None
    rgbs = torch.empty((n, 3, h, w), dtype=torch.float32, device=z.device)
    segm = torch.empty((n, 1, h, w), dtype=torch.float32, device=z.device)
    check(_lib.lib().lp_head_fwd(z.data_ptr(), _p(t), rgbs.data_ptr(), segm.data_ptr(), n, h, w, _stream()), 'lp_head_fwd')
    return t, rgbs, segm


def head_bwd(t: Tensor, d_rgbs: Tensor, d_segm: Optional[Tensor]) -> Tensor:
    _chk(t, 't'); _chk(d_rgbs, 'd_rgbs')
    n, h, w, _ = t.shape
    dz = torch.empty_like(t)
    if d_segm is not None:
        _chk(d_segm, 'd_segm')
    check(_lib.lib().lp_head_bwd(t.data_ptr(), d_rgbs.data_ptr(), _p(d_segm), dz.data_ptr(), n, h, w, _stream()), 'lp_head_bwd')
    return dz


def relu_bwd(dA: Tensor, x: Tensor) -> Tensor:
    _chk(dA, 'dA'); _chk(x, 'x')
    dx = torch.empty_like(x)
    check(_lib.lib().lp_relu_bwd(dA.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), _stream()), 'lp_relu_bwd')
    return dx


def avgpool2_fwd(x: Tensor, relu_in: bool) -> Tensor:
    _chk(x, 'x')
    n, h2, w2, c = x.shape
    y = torch.empty((n, h2 // 2, w2 // 2, c), dtype=torch.float32, device=x.device)
    check(_lib.lib().lp_avgpool2_fwd(x.data_ptr(), y.data_ptr(), n, h2 // 2, w2 // 2, c, int(relu_in), _stream()), 'lp_avgpool2_fwd')
    return y


def avgpool2_bwd(dy: Tensor, x: Tensor, relu_in: bool) -> Tensor:
    _chk(dy, 'dy'); _chk(x, 'x')
    n, h, w, c = x.shape
    dx = torch.empty_like(x)
    check(_lib.lib().lp_avgpool2_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), n, h, w, c, int(relu_in), _stream()), 'lp_avgpool2_bwd')
    return dx


def l1_sum(a: Tensor, b: Tensor, relu_in: bool, coef: float = 1.0) -> Tensor:
    """coef * sum |relu?(a) - relu?(b)| as a 0-d tensor (block partials + a one-block finalize launch)"""
    _chk(a, 'a'); _chk(b, 'b')
    assert a.shape == b.shape
    buf = torch.empty(_lib.lib().lp_l1_partial_blocks() + 1, dtype=torch.float32, device=a.device)
    out = buf[-1:]
    check(_lib.lib().lp_l1_fwd(a.data_ptr(), b.data_ptr(), buf.data_ptr(), a.numel(), int(relu_in), float(coef), out.data_ptr(),
                               _stream()), 'lp_l1_fwd')
    return out.reshape(())


def l1_bwd(a: Tensor, b: Tensor, grad_out: Tensor, coef: float, relu_in: bool, add: Optional[Tensor] = None) -> Tensor:
    """gradient of coef * sum|relu?(a) - relu?(b)| w.r.t. a, times grad_out; ``add`` (same shape) is summed in"""
    _chk(a, 'a'); _chk(b, 'b')
    if add is not None:
        _chk(add, 'add'); assert add.shape == a.shape
    g = grad_out.reshape(1).contiguous().float()
    da = torch.empty_like(a)
    check(_lib.lib().lp_l1_bwd(a.data_ptr(), b.data_ptr(), g.data_ptr(), float(coef), _p(add), da.data_ptr(), a.numel(), int(relu_in), _stream()),
          'lp_l1_bwd')
    return da
