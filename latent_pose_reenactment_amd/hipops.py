"""Tensor-level wrappers around the C ABI.  Activations are NHWC fp32 contiguous CUDA tensors of shape [N,H,W,C]; the operands of
the conv kernels are 16-bit planes (``Act16`` from ``act_pack`` / a conv epilogue, ``WeightPack``).
PyTorch here is plumbing only: it owns the buffers and the stream; all arithmetic happens in liblp_hip.so."""
import os
from typing import NamedTuple, Optional, Tuple

import torch

from . import _lib
from ._lib import PREC_BF16, PREC_BF16X3, PREC_F16, check

Tensor = torch.Tensor


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Optional live profiling hook used by bench.py: when PROFILE is a list, every conv / wgrad launch is bracketed by
# HIP events recorded on the launch stream and appended as (kind, flops, start_event, end_event).
PROFILE = None


class _Timed:
    def __init__(self, kind, flops, tag=None, nbytes=0.0, mm=1):
        """``nbytes``: algorithmic HBM bytes of the launch -- a total, or (read, write); ``mm``: MFMAs per algorithmic MAC (bf16x3: 3)"""
        self.mm = mm
        self.rw = tuple(nbytes) if isinstance(nbytes, tuple) else None
        self.kind, self.flops, self.tag, self.nbytes = kind, flops, tag, (sum(nbytes) if isinstance(nbytes, tuple) else nbytes)

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.e1.record()
            PROFILE.append((self.kind, self.flops, self.e0, self.e1, self.tag, self.nbytes, self.rw, self.mm))


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: Tensor, name: str):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f'{name}: expected a contiguous fp32 CUDA tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}')


def _round_up(v, m):
    return (v + m - 1) // m * m


def _f16(prec: int) -> int:
    return int(prec == PREC_F16)


class WeightPack(NamedTuple):
    hi: Tensor
    lo: Optional[Tensor]
    rows: int      # logical rows (Cout for mode 0, Cin for mode 1)
    cols: int
    rows_p: int
    cols_p: int
    taps: int


def _pack_geometry(w: Tensor, mode: int, small_k: bool):
    """(cout, cin, taps in, taps out, rows, cols, rows_p, cols_p) of a pack: modes 0 / 2 are [taps][Cout][Cin] images, 1 / 3 [taps][Cin][Cout];
    2 / 3 = the phase forms of a x2-upsampled 3x3 conv, 4 / 5 = forward / data gradient of a 3x3 conv followed by AvgPool2d(2) (16 taps:
    lp_pack_weights)"""
    cout, cin = w.shape[0], w.shape[1]
    taps = w.numel() // (cout * cin)
    assert mode in (0, 1) or taps == 9, 'the phase forms (modes 2 / 3) pack a 3x3 weight'
    rows, cols = (cout, cin) if mode in (0, 2, 4) else (cin, cout)
    rows_p = _round_up(rows, 128)
    cols_p = _round_up(cols, 32 if (small_k and cols <= 32) else 64)
    return cout, cin, taps, (16 if mode >= 2 else taps), rows, cols, rows_p, cols_p


def pack_weights(w: Tensor, mode: int, prec: int, small_k: bool = False) -> WeightPack:
    """w: [Cout, Cin, k, k] (or [Cout, Cin]) fp32.  mode 0 = forward pack, mode 1 = dgrad pack (flipped + transposed); 2 / 3 = the phase forms
    (forward / data gradient) of a x2-upsampled 3x3 conv.  small_k pads the contraction dim to 32 instead of 64 (only valid for 3x3
    non-upsampled convs)."""
    _chk(w, 'w')
    cout, cin, taps, taps_out, rows, cols, rows_p, cols_p = _pack_geometry(w, mode, small_k)
    hi = torch.empty((taps_out, rows_p, cols_p), dtype=torch.int16, device=w.device)
    lo = torch.empty_like(hi) if prec == PREC_BF16X3 else None
    check(_lib.lib().lp_pack_weights(w.data_ptr(), hi.data_ptr(), _p(lo), cout, cin, taps, rows_p, cols_p, mode, _f16(prec), _stream()),
          'lp_pack_weights')
    return WeightPack(hi, lo, rows, cols, rows_p, cols_p, taps_out)


def phase_weights(w: Tensor) -> Tensor:
    """w [Cout, Cin, 3, 3] -> [Cout, Cin, 4, 2, 2]: the phase form of conv3x3(nearest_up2(x)).  Output pixel (2y + a, 2x + b) reads the upsampled rows
    2y + a + dy - 1, i.e. the low-resolution rows y - 1, y, y (a = 0) / y, y, y + 1 (a = 1) for dy = 0, 1, 2: taps that meet the same low-resolution
    pixel are summed -- rows {0}, {1, 2} for a = 0 and {0, 1}, {2} for a = 1 (columns alike); tap (i, j) of phase (a, b) reads pixel
    (y + a - 1 + i, x + b - 1 + j).  Plain fp32 sums of the reference's taps (generators/common/blocks.py:74-88)."""
    r = [[w[:, :, 0], w[:, :, 1] + w[:, :, 2]], [w[:, :, 0] + w[:, :, 1], w[:, :, 2]]]          # [a][i] -> [Cout, Cin, 3 (dx)]
    out = []
    for a in range(2):
        for b in range(2):
            taps = []
            for i in range(2):
                row = r[a][i]
                cols = [[row[..., 0], row[..., 1] + row[..., 2]], [row[..., 0] + row[..., 1], row[..., 2]]][b]
                taps.append(torch.stack(cols, dim=-1))
            out.append(torch.stack(taps, dim=-2))          # [Cout, Cin, 2 (i), 2 (j)]
    return torch.stack(out, dim=2).contiguous()           # [Cout, Cin, 4 (phase = 2a + b), 2, 2]


def pack_phase_weights(w: Tensor, prec: int, dgrad: bool = False) -> WeightPack:
    """[4 phases x 4 taps] pack of the phase-decomposed x2-upsampled 3x3 conv: forward [CoutP][CinP] (``conv16(..., upsample=True, phase=True)``) or
    data gradient [CinP][CoutP] -- lp_pack_weights modes 2 / 3 (the tap sums happen inside the pack launch; ``phase_weights`` is their torch form)"""
    return pack_weights(w, 3 if dgrad else 2, prec)


class PackBatch:
    """16-bit packs of MANY conv weights, refreshed by one or two launches.  ``specs`` = [(w, mode, small_k)]; the pack buffers and
    the device descriptor tables are allocated once (static addresses: hipGraph friendly); ``update()`` re-packs all entries from
    the current weight values and returns the list of WeightPacks (same order as ``specs``).  A weight that appears in both
    orientations (training: forward + data-gradient pack) is handled by lp_pack_weights_pairs -- one coalesced read of W through an
    LDS tile for both packs; the rest by lp_pack_weights_batch.  ``precs``: operand mode per entry (a net whose blocks run in different
    modes: nn.Generator's fp16 tail); default: ``prec`` for all."""

    def __init__(self, specs, prec: int, precs=None):
        import struct
        assert _lib.lib().lp_pack_desc_bytes() == 56 and _lib.lib().lp_pack_pair_desc_bytes() == 80
        self.prec = prec
        self.precs = tuple(precs) if precs is not None else (prec,) * len(specs)
        assert len(self.precs) == len(specs)
        self.key = tuple((w.data_ptr(), mode, bool(sk)) for w, mode, sk in specs)
        self.packs = []
        geo = []
        for (w, mode, small_k), prec in zip(specs, self.precs):
            _chk(w, 'w')
            cout, cin, taps, taps_out, rows, cols, rows_p, cols_p = _pack_geometry(w, mode, small_k)
            hi = torch.empty((taps_out, rows_p, cols_p), dtype=torch.int16, device=w.device)
            lo = torch.empty_like(hi) if prec == PREC_BF16X3 else None
            self.packs.append(WeightPack(hi, lo, rows, cols, rows_p, cols_p, taps_out))
            geo.append((w, mode, cout, cin, taps))
        by_w = {}
        for i, (w, mode, *_rest) in enumerate(geo):
            by_w.setdefault(w.data_ptr(), {}).setdefault(mode, i)
        paired = {}
        for ptr, modes in by_w.items():
            if 0 in modes and 1 in modes and geo[modes[0]][4] in (1, 9):
                paired[modes[0]] = modes[1]
        in_pair = set(paired) | set(paired.values())
        blob, pblob = bytearray(), bytearray()
        self.chunks = self.tiles = self.nsingle = self.npair = 0
        lop = lambda pk: 0 if pk.lo is None else pk.lo.data_ptr()
        for i, (w, mode, cout, cin, taps) in enumerate(geo):
            pk = self.packs[i]
            prec = self.precs[i]
            if i in paired:
                pk1 = self.packs[paired[i]]
                assert self.precs[paired[i]] == prec, 'both orientations of one weight are packed in one operand mode'
                tco = (max(pk.rows_p, pk1.cols_p) + 31) // 32
                tci = (max(pk.cols_p, pk1.rows_p) + 31) // 32
                pblob += struct.pack('<QQQQQiiiiiiiiii', w.data_ptr(), pk.hi.data_ptr(), lop(pk), pk1.hi.data_ptr(), lop(pk1), cout, cin, taps,
                                     pk.rows_p, pk.cols_p, pk1.rows_p, pk1.cols_p, self.tiles, tci, _f16(prec))
                self.tiles += tco * tci
                self.npair += 1
            elif i not in in_pair:
                blob += struct.pack('<QQQiiiiiiii', w.data_ptr(), pk.hi.data_ptr(), lop(pk), cout, cin, taps, pk.rows_p, pk.cols_p, mode,
                                    self.chunks, _f16(prec))
                self.chunks += (pk.taps * pk.rows_p * pk.cols_p + 1023) // 1024
                self.nsingle += 1
        dev = specs[0][0].device
        self.table = torch.frombuffer(blob, dtype=torch.uint8).clone().to(dev) if self.nsingle else None
        self.ptable = torch.frombuffer(pblob, dtype=torch.uint8).clone().to(dev) if self.npair else None

    def update(self):
        if self.npair:
            check(_lib.lib().lp_pack_weights_pairs(self.ptable.data_ptr(), self.npair, self.tiles, _stream()), 'lp_pack_weights_pairs')
        if self.nsingle:
            check(_lib.lib().lp_pack_weights_batch(self.table.data_ptr(), self.nsingle, self.chunks, _stream()), 'lp_pack_weights_batch')
        return self.packs


class Act16(NamedTuple):
    """Operand planes of a conv input: ``hi`` (, ``lo``) int16 [N,H,W,C8] holding act(x) * scale in the operand format of the
    precision mode; ``c`` = logical channels (C8 = c rounded up to 8, pad channels zero); ``inv`` = device scalar 1/scale of an
    fp16 gradient operand (None: unscaled)."""
    hi: Tensor
    lo: Optional[Tensor]
    c: int
    inv: Optional[Tensor]

    @property
    def nhw(self):
        return tuple(self.hi.shape[:3])


def _alloc16(n, h, w, c, prec, device):
    c8 = _round_up(c, 8)
    hi = torch.empty((n, h, w, c8), dtype=torch.int16, device=device)
    lo = torch.empty_like(hi) if prec == PREC_BF16X3 else None
    return hi, lo


FUSE_AMAX = os.environ.get('LP_FUSE_AMAX', '1') != '0'      # LP_FUSE_AMAX=0: every gradient operand gets its own amax pass (test knob)
AMAX_STATS = {'fused': 0, 'pass': 0}


class _AmaxSlots:
    """Zeroed slot groups (lp_amax_slots() slots, lp_amax_slot_stride() floats apart) for the kernels that fold max|y| of a gradient tensor into their epilogue.
    Handed out from one buffer zeroed by a single fill launch; under hipGraph capture the buffer (and its fill) belong to the
    capture, so every replay starts from zeros."""

    def __init__(self):
        self.buf, self.pos, self.captured, self.n = None, 0, False, 0

    def take(self, device):
        if not self.n:
            self.n = _lib.lib().lp_amax_slots() * _lib.lib().lp_amax_slot_stride()
        capturing = torch.cuda.is_current_stream_capturing()
        if (self.buf is None or self.pos + self.n > self.buf.numel() or self.buf.device != device or capturing != self.captured):
            self.buf = torch.zeros(self.n * 256, dtype=torch.float32, device=device)
            self.pos, self.captured = 0, capturing
        s = self.buf[self.pos:self.pos + self.n]
        self.pos += self.n
        return s


_AMAX_SLOTS = _AmaxSlots()


def _amax_attach(out: Tensor, want: bool):
    """slot group for a kernel about to write gradient tensor ``out`` (or None): remembered on the tensor object, together with its
    version counter -- an in-place update (autograd accumulating a second gradient into it) invalidates the recorded maximum"""
    if not (want and FUSE_AMAX):
        return None
    slots = _AMAX_SLOTS.take(out.device)
    out._lp_amax = (slots, out._version)
    return slots


def act_pack(x: Tensor, *, pro: int = 0, scale: Optional[Tensor] = None, shift: Optional[Tensor] = None, prec: int = PREC_BF16,
             grad: bool = False) -> Act16:
    """x [N,H,W,C] fp32 -> operand planes of act(x): pro 0 identity | 1 relu(x*scale[n,c]+shift[n,c]) | 2 relu(x) |
    3 relu6(x*scale[c]+shift[c]).
    ``grad``: x is a gradient (dY): in fp16 mode it is scaled by a power of two taken from its amax (two extra small launches) so
    that it sits in the fp16 normal range; the consumer gets 1/scale through ``Act16.inv``."""
    _chk(x, 'x')
    n, h, w, c = x.shape
    hi, lo = _alloc16(n, h, w, c, prec, x.device)
    sc = part = None
    npart, pstride = 0, 1
    if grad and prec == PREC_F16:
        rec = getattr(x, '_lp_amax', None)
        if rec is not None and rec[1] == x._version and FUSE_AMAX:
            part, npart, pstride = rec[0], _lib.lib().lp_amax_slots(), _lib.lib().lp_amax_slot_stride()    # folded in by the kernel that wrote x
            sc = torch.empty(2, dtype=torch.float32, device=x.device)
            AMAX_STATS['fused'] += 1
        else:
            npart = _lib.lib().lp_amax_blocks()
            buf = torch.empty(npart + 2, dtype=torch.float32, device=x.device)
            part, sc = buf[:npart], buf[npart:]
            check(_lib.lib().lp_amax_partial(x.data_ptr(), x.numel(), part.data_ptr(), _stream()), 'lp_amax_partial')
            AMAX_STATS['pass'] += 1
    for t, nm in ((scale, 'scale'), (shift, 'shift')):
        if t is not None:
            _chk(t, nm)
    check(_lib.lib().lp_act_pack(x.data_ptr(), _p(scale), _p(shift), pro, hi.data_ptr(), _p(lo), n, h * w, c, prec, None, _p(part), npart,
                                 pstride, _p(sc), _stream()), 'lp_act_pack')
    return Act16(hi, lo, c, None if sc is None else sc[1:])


class ConvStats(NamedTuple):
    """{count, mean, M2} partials a conv epilogue left behind: ``part`` [rows, Cout, 3] floats, ``rows`` partial rows per image"""
    part: Tensor
    rows: int
    images: int = 1         # images of the launch: a BatchNorm over all of them merges rows * images partials per channel


def conv16(a: Act16, pack: WeightPack, *, ksize: int, upsample: bool = False, bias: Optional[Tensor] = None,
           res: Optional[Tensor] = None, res_shift: int = 0, alpha: Optional[Tensor] = None, prec: int = PREC_BF16,
           relu_mask: Optional[Act16] = None, out16: Optional[int] = None, amax: bool = False, stats: bool = False, want_y: bool = True,
           kind: str = 'conv_igemm', phase: bool = False, phase_dgrad: bool = False, y_relu: bool = False):
    """y = alpha * conv(up2?(a), pack) + bias + res on operand planes; a [N,Hin,Win,C8] -> y [N,H,W,Cout] fp32.
    ``relu_mask``: operand planes [N,H,W,Co8] of the forward conv's input; y is zeroed where they are <= 0 (fused ReLU backward
    when this launch is a data gradient).  ``out16`` = 0 | 1: also return the operand planes of y (1: of relu(y)) -> (y, Act16).
    ``stats``: the epilogue also leaves the norm-statistics partials of y -> (..., ConvStats | None) appended (None: geometry not covered
    by the fused path -- run ``instnorm_stats`` / ``bn_train_stats`` on y).  ``want_y=False`` (with ``out16`` and Cout % 8 == 0): no fp32 y
    is written, y is returned as None.  ``phase`` (with ``upsample``, 3x3; round 6): ``pack`` is a ``pack_phase_weights`` image and the conv runs in
    its phase-decomposed form -- per output phase (a, b) a 2 x 2 conv on the low-resolution planes with the coinciding taps pre-summed: the same
    result with 4/9 of the matrix work.  ``phase_dgrad``: ``a`` = the planes of dy [N, 2H, 2W, C8] of such a conv, ``pack`` =
    ``pack_phase_weights(w, prec, dgrad=True)`` -> the gradient w.r.t. the LOW-resolution input [N, H, W, Cout] (the upsample's 2x2 sum included).
    The same two kernel forms run a 3x3 conv FOLLOWED BY AvgPool2d(2) as one 4x4 stride-2 conv (``pack_weights`` modes 4 / 5; nn.ConvPoolFn):
    ``phase_dgrad`` with a mode-4 pack = pooled conv output from the full-resolution planes, ``phase`` with a mode-5 pack = its data gradient.
    ``y_relu``: y is stored as relu(y) (the next block's in-place ReLU)."""
    n, hin, win = a.nhw
    cin = a.c
    assert not phase or (upsample and ksize == 3)
    assert not phase_dgrad or (ksize == 3 and not upsample and not phase and hin % 2 == 0 and win % 2 == 0)
    assert cin == pack.cols and pack.taps == (16 if (phase or phase_dgrad) else ksize * ksize), (a.hi.shape, a.c, pack.rows, pack.cols, pack.taps)
    h, w = (hin * 2, win * 2) if upsample else (hin // 2, win // 2) if phase_dgrad else (hin, win)
    cout = pack.rows
    dev = a.hi.device
    if not want_y:
        assert out16 is not None and cout % 8 == 0 and not amax, 'want_y=False needs out16 and Cout % 8 == 0'
    y = torch.empty((n, h, w, cout), dtype=torch.float32, device=dev) if want_y else None
    for t, nm in ((bias, 'bias'), (res, 'res')):
        if t is not None:
            _chk(t, nm)
    if res is not None:
        assert res.shape == (n, h >> res_shift, w >> res_shift, cout), (res.shape, (n, h, w, cout), res_shift)
    if relu_mask is not None:
        assert relu_mask.hi.shape == (n, h, w, _round_up(cout, 8)), (relu_mask.hi.shape, (n, h, w, cout))
    o_hi = o_lo = None
    if out16 is not None:
        o_hi, o_lo = _alloc16(n, h, w, cout, prec, dev)
    slots = _amax_attach(y, amax and prec == PREC_F16) if y is not None else None       # y is a gradient that will be packed: max|y| from the epilogue
    ws_bytes = _lib.lib().lp_conv16_fwd_workspace_bytes(n, h, w, cout, ksize)          # split-K partial tiles (small feature maps)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev) if ws_bytes else None
    st_buf, st_rows, st_cap = None, None, 0
    if stats:
        import ctypes
        st_cap = _lib.lib().lp_conv16_stats_floats(n, h, w, cout)
        st_buf = torch.empty(st_cap, dtype=torch.float32, device=dev)
        st_rows = ctypes.c_int(0)
    pl = 4 if prec == PREC_BF16X3 else 2              # bytes per operand-plane element
    nbytes = (n * hin * win * a.hi.shape[3] * pl + pack.hi.numel() * pl + (n * (h >> res_shift) * (w >> res_shift) * cout * 4 if res is not None else 0),
              n * h * w * cout * ((4 if want_y else 0) + (pl if out16 is not None else 0)))          # (read: planes once + weights + residual, write)
    # (roofline accounting: the DENSE count of the conv as the reference executes it -- the phase forms do 4/9 of it; the data-gradient form's dense
    #  conv lives on the 2h x 2w grid)
    dh, dw_ = (2 * h, 2 * w) if phase_dgrad else (h, w)
    with _Timed(kind, 2.0 * n * dh * dw_ * cout * cin * ksize * ksize, (n, dh, dw_, cin, cout, ksize, 3 if phase_dgrad else 2 if phase else int(upsample), 0), nbytes,
                mm=3 if prec == PREC_BF16X3 else 1):
        check(_lib.lib().lp_conv16_fwd_stats(a.hi.data_ptr(), _p(a.lo), pack.hi.data_ptr(), _p(pack.lo), _p(y), _p(bias), _p(res),
                                             _p(alpha), _p(a.inv), n, h, w, cin, cout, pack.cols_p, pack.rows_p, ksize, 3 if phase_dgrad else 2 if phase else int(upsample),
                                             res_shift, prec, None if relu_mask is None else relu_mask.hi.data_ptr(), _p(o_hi), _p(o_lo),
                                             int(bool(out16)) | (2 if y_relu else 0), _p(ws), ws_bytes, _p(slots), _p(st_buf), st_cap,
                                             None if st_rows is None else ctypes.addressof(st_rows), _stream()), 'lp_conv16_fwd')
    out = (y,) if out16 is None else (y, Act16(o_hi, o_lo, cout, None))
    if stats:
        out = out + ((ConvStats(st_buf, st_rows.value, n) if st_rows.value > 0 else None),)
    return out[0] if len(out) == 1 else out


_THIN = os.environ.get('LP_THIN', '1') != '0'      # LP_THIN=0: thin-channel layers through the MFMA kernels (test knob)
_THIN_MFMA = os.environ.get('LP_THIN_MFMA', '1') != '0'


def thin_conv_supported(cin: int, cout: int, ksize: int, w: int) -> bool:
    return _THIN and bool(_lib.lib().lp_thin_conv_supported(cin, cout, ksize, w))


def thin_conv(x: Tensor, pack: WeightPack, *, ksize: int, bias: Optional[Tensor] = None, alpha: Optional[Tensor] = None,
              prec: int = PREC_BF16, out16: Optional[int] = None, out16_prec: Optional[int] = None, want_y: bool = True):
    """conv with <= 4 input channels (RGB -> 64, dz -> 64) on the plain NHWC tensor (no input operand planes): fp32 VALU kernel, or --
    RGB 3x3 in the fp16 / bf16 modes -- one MFMA k-step per output block.  ``out16`` = 0 | 1: also return the operand planes of y
    (1: of relu(y)) -> (y, Act16); written by the same launch where the MFMA kernel runs, by a pack pass otherwise."""
    _chk(x, 'x')
    n, h, w, cin = x.shape
    cout = pack.rows
    # ``out16_prec``: operand mode of the emitted planes when it differs from this conv's own (a strict first layer feeding fp16 layers)
    oprec = prec if out16_prec is None else out16_prec
    fused = out16 is not None and oprec == prec and bool(_lib.lib().lp_thin_conv_emits_planes(cin, cout, ksize, w, prec)) and _THIN_MFMA
    # ``want_y=False``: planes-only output where the MFMA kernel emits them (else y is produced and dropped by the caller)
    y = torch.empty((n, h, w, cout), dtype=torch.float32, device=x.device) if (want_y or not fused) else None
    o_hi = torch.empty((n, h, w, cout), dtype=torch.int16, device=x.device) if fused else None
    with _Timed('conv_thin', 2.0 * n * h * w * cout * cin * ksize * ksize, (n, h, w, cin, cout, ksize, 0, 0)):
        check(_lib.lib().lp_thin_conv_fwd(x.data_ptr(), pack.hi.data_ptr(), _p(pack.lo), _p(y), _p(bias), _p(alpha), n, h, w, cin,
                                          cout, pack.cols_p, pack.rows_p, ksize, prec, _p(o_hi), int(bool(out16)), _stream()),
              'lp_thin_conv_fwd')
    if out16 is None:
        return y
    if fused:
        return y, Act16(o_hi, None, cout, None)
    return y, act_pack(y, pro=2 if out16 else 0, prec=oprec)


def conv(x: Tensor, pack: WeightPack, *, ksize: int, upsample: bool = False, pro: int = 0, scale: Optional[Tensor] = None,
         shift: Optional[Tensor] = None, bias: Optional[Tensor] = None, res: Optional[Tensor] = None, res_shift: int = 0,
         alpha: Optional[Tensor] = None, prec: int = PREC_BF16, relu_mask: Optional[Tensor] = None, grad: bool = False) -> Tensor:
    """Convenience form on an fp32 NHWC tensor: y = alpha * conv(up2?(act(x)), pack) + bias + res = lp_act_pack + lp_conv16_fwd
    (or the thin-channel kernel for <= 4 input channels).  ``relu_mask`` fp32 [N,H,W,Cout]: y zeroed where relu_mask <= 0.
    ``grad``: x is a gradient (fp16 mode scales it).  Callers that reuse the planes (forward + weight gradient) use
    ``act_pack`` / ``conv16`` directly."""
    _chk(x, 'x')
    cin = x.shape[3]
    if (not upsample and pro == 0 and res is None and relu_mask is None and thin_conv_supported(cin, pack.rows, ksize, x.shape[2])):
        return thin_conv(x, pack, ksize=ksize, bias=bias, alpha=alpha, prec=prec)
    a = act_pack(x, pro=pro, scale=scale, shift=shift, prec=prec, grad=grad)
    m16 = None
    if relu_mask is not None:
        m16 = act_pack(relu_mask, pro=0, prec=PREC_BF16)      # (bf16 keeps the sign / zero-ness of every normal fp32 value)
    return conv16(a, pack, ksize=ksize, upsample=upsample, bias=bias, res=res, res_shift=res_shift, alpha=alpha, prec=prec, relu_mask=m16)


def default_splits(n, h, w, cin, cout, ksize=3, prec=None, bias=False):
    """pixel-range splits of the weight-gradient launch.  3x3: ~2 workgroups per CU in total (9 taps of MFMA work per staged tile).
    1x1 without a bias gradient (wgrad1x1_kernel: 128 x 128 weight tiles, 64-pixel stages): one resident set of workgroups -- 2 per CU
    (64 KB of LDS each; 1 per CU for bf16x3's doubled planes) -- each walking >= 8 stages, a multiple of 8 so that the tiles of one
    pixel range share an XCD.  1x1 with a bias gradient (conv_wgrad_kernel): ~4 workgroups per CU, >= 4 128-pixel tiles each."""
    if ksize == 1 and not bias and os.environ.get('LP_WGRAD1X1_OLD') is None:
        stages = (n * h * w + 63) // 64
        blocks = (_round_up(cout, 64) + 127) // 128 * ((_round_up(cin, 64) + 127) // 128)
        s = max(1, min(512 // blocks, stages // 8))          # (bf16x3 too since round 4: 32-pixel stages, two workgroups per CU)
        return s // 8 * 8 if s >= 8 else s
    tiles = (n * h * w + 127) // 128
    if ksize == 1:
        blocks = (_round_up(cout, 128) // 128 if cout >= 128 else 1) * (_round_up(cin, 64) // 64)
        return max(1, min(1024 // blocks, tiles // 4))
    blocks = _round_up(cout, 64) // 64 * (_round_up(cin, 64) // 64)
    return max(1, min(512 // blocks, tiles))


# Deferred weight gradients (round 6, one-GPU meta-training step): a weight gradient whose result is ADDED to the parameter's .grad (spectral-norm rule
# included) has no consumer inside the backward pass, so its launch need not sit in the data-gradient chain.  Inside ``wgrad_defer()`` such launches
# are recorded instead of issued; ``wgrad_flush()`` issues them -- on whatever stream is current THEN (runners/holycow.py: the critic-backward
# stream, beside the encoders' backward).  The operand planes stay alive in the job list, which the caller keeps until the flushing stream has been
# joined (the planes were allocated on the backward pass's stream).  Same kernels, same operands, one contribution per parameter: bit-identical sums.
_WG_DEFER = {'active': False, 'jobs': []}


class wgrad_defer:
    def __enter__(self):
        self.prev = _WG_DEFER['active']
        _WG_DEFER['active'] = True

    def __exit__(self, *exc):
        _WG_DEFER['active'] = self.prev
        if exc[0] is not None:
            _WG_DEFER['jobs'] = []


def wgrad_flush():
    """issue the recorded weight-gradient launches on the current stream (inside nn.fused_grad_accumulation: their spectral-norm rules join its
    batched launch) -> the job list (operands): keep it until this stream has been joined"""
    jobs, _WG_DEFER['jobs'] = _WG_DEFER['jobs'], []
    prev, _WG_DEFER['active'] = _WG_DEFER['active'], False
    try:
        for args, kw in jobs:
            out = conv_wgrad16(*args, **kw)
            assert out is None or all(o is None for o in out), 'a deferred weight gradient accumulates: nothing is returned'
    finally:
        _WG_DEFER['active'] = prev
    return jobs


def conv_wgrad16(a: Act16, dy: Act16, *, ksize: int, upsample: bool = False, prec: int = PREC_BF16, splits: Optional[int] = None,
                 sn=None, accum: Optional[Tensor] = None, bias_grad: bool = False, bias_accum: Optional[Tensor] = None, kind: str = 'conv_wgrad'):
    """dw [Cout,Cin,k,k] = sum_pixels dy (x) up2?(a) (shifted by tap) on operand planes (a = what the forward conv consumed).
    ``sn`` = (w_orig, u, v, sig): the layer is spectrally normalised (forward used alpha = 1/sigma in the conv epilogue); the
    returned gradient is then w.r.t. W_orig: dw/sigma - <dw, W_orig>/sigma^2 u v^T.
    ``accum`` (with ``sn``): add that gradient to this tensor (the parameter's .grad) instead and return None.
    ``bias_grad``: return ``(dw, db)`` with db [Cout] = sum_pixels dy, produced by the same launch (the kernel streams dy anyway);
    ``bias_accum`` (the bias parameter's .grad): the launch adds db to it instead and db is returned as None."""
    n, h, w = dy.nhw
    cout, cin = dy.c, a.c
    assert a.nhw == ((n, h // 2, w // 2) if upsample else (n, h, w)), (a.hi.shape, dy.hi.shape, upsample)
    if (_WG_DEFER['active'] and sn is not None and accum is not None and (not bias_grad or bias_accum is not None)
            and SN_DEFER and _SN_DEFER['depth'] > 0):
        _WG_DEFER['jobs'].append(((a, dy), dict(ksize=ksize, upsample=upsample, prec=prec, splits=splits, sn=sn, accum=accum, bias_grad=bias_grad,
                                                bias_accum=bias_accum, kind=kind)))
        return (None, None) if bias_grad else None
    dev = dy.hi.device
    if splits is None:
        splits = default_splits(n, h, w, cin, cout, ksize, prec, bias_grad)
    ws = torch.empty(_lib.lib().lp_conv_wgrad_workspace_bytes(cin, cout, ksize, splits) // 4, dtype=torch.float32, device=dev)
    dw = torch.empty((cout, cin, ksize, ksize), dtype=torch.float32, device=dev)
    db = None
    if bias_grad:
        if bias_accum is not None:
            assert bias_accum.is_contiguous() and bias_accum.dtype == torch.float32 and bias_accum.numel() == cout
            db = bias_accum
        else:
            db = torch.empty(cout, dtype=torch.float32, device=dev)
    dot, ndot = None, 0
    if sn is not None:          # the reduction launch also takes <dw, W_orig> (per-block partials) for lp_sn_grad_apply
        _chk(sn[0], 'w_orig')
        ndot = _lib.lib().lp_conv_wgrad_dot_blocks(cin, cout, ksize)
        dot = torch.empty(ndot, dtype=torch.float32, device=dev)
    pl = 4 if prec == PREC_BF16X3 else 2
    nbytes = (a.hi.numel() + dy.hi.numel()) * pl + cout * cin * ksize * ksize * 4
    with _Timed(kind, 2.0 * n * h * w * cout * cin * ksize * ksize, (n, h, w, cin, cout, ksize, int(upsample), 0), nbytes, mm=3 if prec == PREC_BF16X3 else 1):
        check(_lib.lib().lp_conv16_wgrad(a.hi.data_ptr(), _p(a.lo), dy.hi.data_ptr(), _p(dy.lo), dw.data_ptr(), ws.data_ptr(), n, h, w, cin,
                                         cout, ksize, int(upsample), splits, prec, _p(db), int(bias_grad and bias_accum is not None), _p(dy.inv),
                                         None if sn is None else sn[0].data_ptr(), _p(dot), _stream()), 'lp_conv16_wgrad')
    dw = _sn_finish(dw, sn, accum, dot, ndot)
    return (dw, None if bias_accum is not None else db) if bias_grad else dw


# Launch diet (round 5): inside ``nn.fused_grad_accumulation`` the spectral-norm gradient rule of a conv weight whose result is ACCUMULATED into
# the parameter's .grad (nothing is returned to autograd) is not launched per layer; the jobs are collected and run by lp_sn_grad_apply_batch
# when the context is left (after every side stream is joined, before anything reads .grad).  LP_SN_DEFER=0: per-layer launches.
SN_DEFER = os.environ.get('LP_SN_DEFER', '1') != '0'
_SN_DEFER = {'depth': 0, 'jobs': []}


def sn_defer_begin():
    _SN_DEFER['depth'] += 1


def sn_defer_check(who: str):
    """ a reader of ``.grad`` -- the gradient exchange, an optimizer step -- must not run while deferred spectral-norm jobs are
    pending: the conv weights' gradients are incomplete until ``nn.fused_grad_accumulation`` is left.  Raises instead of reading them early."""
    if _WG_DEFER['jobs']:
        raise RuntimeError(f'{who}: {len(_WG_DEFER["jobs"])} deferred weight-gradient launches are pending (hipops.wgrad_defer without wgrad_flush)')
    if _SN_DEFER['jobs']:
        raise RuntimeError(f'{who}: {len(_SN_DEFER["jobs"])} deferred spectral-norm gradient jobs are pending -- .grad of the conv weights is '
                           'incomplete inside nn.fused_grad_accumulation; leave the context (or set LP_SN_DEFER=0) before reading gradients')


def sn_defer_end(discard: bool = False):
    """leave one level of deferral; the outermost level runs (or, after a failed backward, drops) the collected jobs"""
    _SN_DEFER['depth'] -= 1
    if _SN_DEFER['depth'] > 0:
        return
    jobs, _SN_DEFER['jobs'] = _SN_DEFER['jobs'], []
    if not jobs or discard:
        return
    import ctypes
    import struct
    assert _lib.lib().lp_sn_apply_desc_bytes() == 64
    # jobs that accumulate into the SAME tensor (two passes of the critic depositing on one parameter from one stream) must not share a launch:
    # a job joins the first round that does not hold its target yet; the rounds run one after the other on this stream
    rounds = []
    for job in jobs:
        key = job[6].data_ptr()
        for targets, members in rounds:
            if key not in targets:
                targets.add(key); members.append(job)
                break
        else:
            rounds.append(({key}, [job]))
    for _, members in rounds:
        blob = bytearray()
        for dw, u, v, sig, dot, ndot, accum in members:
            blob += struct.pack('<QQQQQQiiii', dw.data_ptr(), u.data_ptr(), v.data_ptr(), sig.data_ptr(), dot.data_ptr(), accum.data_ptr(), ndot,
                                dw.shape[0], dw.numel() // dw.shape[0], 0)
        buf = (ctypes.c_char * len(blob)).from_buffer(blob)
        check(_lib.lib().lp_sn_grad_apply_batch(ctypes.addressof(buf), len(members), _stream()), 'lp_sn_grad_apply_batch')
    # (the temporaries in `jobs` are released here, AFTER the launch was enqueued on the stream that allocated them or was joined with it)


def _sn_finish(dw, sn, accum, dot=None, ndot=0):
    if sn is None:
        return dw
    w_orig, u, v, sig = sn
    cout = dw.shape[0]
    if accum is not None and ndot > 0 and dot is not None and SN_DEFER and _SN_DEFER['depth'] > 0:
        assert accum.is_contiguous() and accum.dtype == torch.float32 and accum.numel() == dw.numel()
        _SN_DEFER['jobs'].append((dw, u, v, sig, dot, ndot, accum))
        return None
    if dot is None:
        dot = torch.empty(512, dtype=torch.float32, device=dw.device)      # per-block partials of <g, W>, taken by lp_sn_grad_apply
    if accum is not None:
        assert accum.is_contiguous() and accum.dtype == torch.float32 and accum.numel() == dw.numel()
    check(_lib.lib().lp_sn_grad_apply(dw.data_ptr(), w_orig.data_ptr(), u.data_ptr(), v.data_ptr(), sig.data_ptr(), dot.data_ptr(), ndot,
                                      _p(accum), cout, dw.numel() // cout, _stream()), 'lp_sn_grad_apply')
    return None if accum is not None else dw


# ---- reflection padding of the 3x3 ResBlock convs as a border correction of the zero-padded conv (csrc/reflect_border.hip) -----------------------
def reflect_border_fwd(a: Act16, w_orig: Tensor, alpha: Optional[Tensor], y: Tensor, *, prec: int, upsample: bool = False) -> Tensor:
    """y [N,H,W,Cout] (the zero-padded conv of up2?(a) with W_orig * alpha) += the taps that leave the image, applied to the mirrored pixels:
    afterwards y is the conv behind nn.ReflectionPad2d(1) (blocks.py:76-88 with --gen_padding / --dis_padding reflection).  In place."""
    _chk(y, 'y'); _chk(w_orig, 'w_orig')
    n, h, w, cout = y.shape
    assert a.inv is None and a.nhw == ((n, h // 2, w // 2) if upsample else (n, h, w)), (a.hi.shape, y.shape, upsample)
    assert tuple(w_orig.shape) == (cout, a.c, 3, 3), (w_orig.shape, cout, a.c)
    check(_lib.lib().lp_reflect_border_fwd(a.hi.data_ptr(), _p(a.lo), prec, n, h, w, a.c, a.hi.shape[3], int(upsample), w_orig.data_ptr(), cout,
                                           _p(alpha), y.data_ptr(), _stream()), 'lp_reflect_border_fwd')
    return y


def reflect_border_dgrad(dy: Tensor, w_orig: Tensor, alpha: Optional[Tensor], dx: Tensor, mask16: Optional[Act16] = None) -> Tensor:
    """dx [N,H,W,Cin] (the zero-padded data gradient) += the transposed border terms; ``mask16`` = planes of relu(x) when the forward applied ReLU
    in its prologue (the main launch masked its result by [x > 0] in its epilogue; the correction is masked the same way).  In place."""
    _chk(dy, 'dy'); _chk(dx, 'dx'); _chk(w_orig, 'w_orig')
    n, h, w, cout = dy.shape
    cin = dx.shape[3]
    assert tuple(dx.shape[:3]) == (n, h, w) and tuple(w_orig.shape) == (cout, cin, 3, 3), (dx.shape, dy.shape, w_orig.shape)
    assert mask16 is None or mask16.nhw == (n, h, w)
    check(_lib.lib().lp_reflect_border_dgrad(dy.data_ptr(), n, h, w, cout, w_orig.data_ptr(), cin, _p(alpha), None if mask16 is None else mask16.hi.data_ptr(),
                                             0 if mask16 is None else mask16.hi.shape[3], dx.data_ptr(), _stream()), 'lp_reflect_border_dgrad')
    return dx


def reflect_border_wgrad(a: Act16, dy: Tensor, *, prec: int, upsample: bool = False, sn=None, accum: Optional[Tensor] = None) -> Optional[Tensor]:
    """the border terms' share of the weight gradient, [Cout,Cin,3,3]; ``sn`` / ``accum`` as in ``conv_wgrad16`` (the spectral-norm gradient rule is
    linear in the raw gradient, so the share is passed through it on its own and added to the same .grad): None when accumulated."""
    _chk(dy, 'dy')
    n, h, w, cout = dy.shape
    assert a.inv is None and a.nhw == ((n, h // 2, w // 2) if upsample else (n, h, w)), (a.hi.shape, dy.shape, upsample)
    gw = torch.empty((cout, a.c, 3, 3), dtype=torch.float32, device=dy.device)
    nws = _lib.lib().lp_reflect_border_wgrad_workspace_bytes(n, h, w, a.c, cout) // 4
    ws = torch.empty(nws, dtype=torch.float32, device=dy.device) if nws else None
    check(_lib.lib().lp_reflect_border_wgrad(a.hi.data_ptr(), _p(a.lo), prec, n, h, w, a.c, a.hi.shape[3], int(upsample), dy.data_ptr(), cout,
                                             gw.data_ptr(), _p(ws), _stream()), 'lp_reflect_border_wgrad')
    return _sn_finish(gw, sn, accum)


def thin_wgrad_supported(cin: int, cout: int, ksize: int, pro: int, w: int) -> bool:
    return _THIN and bool(_lib.lib().lp_thin_wgrad_supported(cin, cout, ksize, pro, w))


def thin_wgrad(x: Tensor, dy: Tensor, *, ksize: int, pro: int = 0, scale: Optional[Tensor] = None, shift: Optional[Tensor] = None,
               splits: Optional[int] = None, sn=None, accum: Optional[Tensor] = None, bias_grad: bool = False):
    """weight gradient of a conv with <= 4 channels on one side, fp32 NHWC tensors in (prologue of the wide input applied on the fly)"""
    _chk(x, 'x'); _chk(dy, 'dy')
    n, h, w, cout = dy.shape
    cin = x.shape[3]
    if splits is None:
        splits = default_splits(n, h, w, cin, cout)
    ws = torch.empty(_lib.lib().lp_conv_wgrad_workspace_bytes(cin, cout, ksize, splits) // 4, dtype=torch.float32, device=x.device)
    dw = torch.empty((cout, cin, ksize, ksize), dtype=torch.float32, device=x.device)
    db = None
    if bias_grad and _lib.lib().lp_thin_wgrad_has_dbias(cin, cout):
        db = torch.empty(cout, dtype=torch.float32, device=x.device)
    with _Timed('wgrad_thin', 2.0 * n * h * w * cout * cin * ksize * ksize, (n, h, w, cin, cout, ksize, 0, pro)):
        check(_lib.lib().lp_thin_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), _p(scale), _p(shift), n, h, w, cin, cout,
                                       ksize, pro, splits, _p(db), _stream()), 'lp_thin_wgrad')
    if bias_grad and db is None:
        db = dy.sum(dim=(0, 1, 2))
    dw = _sn_finish(dw, sn, accum)
    return (dw, db) if bias_grad else dw


def conv_wgrad(x: Tensor, dy: Tensor, *, ksize: int, upsample: bool = False, pro: int = 0, scale: Optional[Tensor] = None,
               shift: Optional[Tensor] = None, prec: int = PREC_BF16, splits: Optional[int] = None, sn=None,
               accum: Optional[Tensor] = None, bias_grad: bool = False):
    """Convenience form on fp32 NHWC tensors: packs act(x) and dy, then lp_conv16_wgrad (or the thin-channel kernel)."""
    _chk(x, 'x'); _chk(dy, 'dy')
    if not upsample and thin_wgrad_supported(x.shape[3], dy.shape[3], ksize, pro, dy.shape[2]):
        return thin_wgrad(x, dy, ksize=ksize, pro=pro, scale=scale, shift=shift, splits=splits, sn=sn, accum=accum, bias_grad=bias_grad)
    a = act_pack(x, pro=pro, scale=scale, shift=shift, prec=prec)
    d = act_pack(dy, pro=0, prec=prec, grad=True)
    return conv_wgrad16(a, d, ksize=ksize, upsample=upsample, prec=prec, splits=splits, sn=sn, accum=accum, bias_grad=bias_grad)


def linear_supported(b: int, k: int, fwd_only: bool = False) -> bool:
    return 1 <= b <= 64 and k % 4 == 0 and k <= (2048 if fwd_only else 1024)


def linear_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor], alpha: Optional[Tensor]) -> Tensor:
    """y [B,N] = alpha * x W^T + bias (lp_linear_fwd: one pass over W, 1/sigma and bias fused)"""
    _chk(x, 'x'); _chk(w, 'w')
    b, k = x.shape
    n = w.shape[0]
    y = torch.empty((b, n), dtype=torch.float32, device=x.device)
    check(_lib.lib().lp_linear_fwd(x.data_ptr(), w.data_ptr(), _p(bias), _p(alpha), y.data_ptr(), b, n, k, _stream()), 'lp_linear_fwd')
    return y


def linear_bwd(x: Tensor, w: Tensor, g: Tensor, alpha: Optional[Tensor], want_dx: bool, want_dw: bool, want_db: bool):
    """-> (dx = alpha * g W | None, raw dw = g^T x | None, db | None) in one pass over W"""
    _chk(x, 'x'); _chk(w, 'w'); _chk(g, 'g')
    b, k = x.shape
    n = w.shape[0]
    dev = x.device
    dx = torch.empty((b, k), dtype=torch.float32, device=dev) if want_dx else None
    dw = torch.empty((n, k), dtype=torch.float32, device=dev) if want_dw else None
    db = torch.empty(n, dtype=torch.float32, device=dev) if want_db else None
    ws = torch.empty(_lib.lib().lp_linear_bwd_workspace_bytes(b, n, k) // 4, dtype=torch.float32, device=dev) if want_dx else None
    check(_lib.lib().lp_linear_bwd(x.data_ptr(), w.data_ptr(), g.data_ptr(), _p(alpha), _p(dx), _p(dw), _p(db), _p(ws), b, n, k, _stream()),
          'lp_linear_bwd')
    return dx, dw, db


def grid_crop_fwd(images: Tensor, boxes: Tensor, out_hw) -> Tensor:
    """images [N,C,H,W] fp32 NCHW, boxes [N,4] (t, b, l, r) -> crops [N,C,Ho,Wo] (bilinear, reflection, align_corners=False)"""
    _chk(images, 'images'); _chk(boxes, 'boxes')
    n, c, h, w = images.shape
    out = torch.empty((n, c) + tuple(out_hw), dtype=torch.float32, device=images.device)
    check(_lib.lib().lp_grid_crop_fwd(images.data_ptr(), boxes.data_ptr(), out.data_ptr(), n, c, h, w, out_hw[0], out_hw[1], _stream()), 'lp_grid_crop_fwd')
    return out


def grid_crop_bwd(dout: Tensor, boxes: Tensor, in_shape) -> Tensor:
    _chk(dout, 'dout'); _chk(boxes, 'boxes')
    n, c, h, w = in_shape
    dimg = torch.empty(in_shape, dtype=torch.float32, device=dout.device)
    check(_lib.lib().lp_grid_crop_bwd(dout.data_ptr(), boxes.data_ptr(), dimg.data_ptr(), n, c, h, w, dout.shape[2], dout.shape[3], _stream()),
          'lp_grid_crop_bwd')
    return dimg


# ---- MobileNetV2 pose encoder (forward): layers that are not dense contractions ------------------------------------------------
def stem_conv_s2(x: Tensor, w: Tensor) -> Tensor:
    """x [N,3,H,W] NCHW fp32, w [Cout,3,3,3] -> y [N,H/2,W/2,Cout] NHWC (3x3, stride 2, pad 1)"""
    _chk(x, 'x'); _chk(w, 'w')
    n, c, h, wd = x.shape
    assert c == 3 and tuple(w.shape[1:]) == (3, 3, 3), (x.shape, w.shape)
    y = torch.empty((n, h // 2, wd // 2, w.shape[0]), dtype=torch.float32, device=x.device)
    check(_lib.lib().lp_stem_conv_s2(x.data_ptr(), w.data_ptr(), y.data_ptr(), n, h, wd, w.shape[0], _stream()), 'lp_stem_conv_s2')
    return y


def dwconv3x3(x: Tensor, w: Tensor, stride: int, in_scale: Optional[Tensor] = None, in_shift: Optional[Tensor] = None) -> Tensor:
    """depthwise 3x3 (pad 1) of relu6(x*in_scale[c]+in_shift[c]) (or of x); x [N,H,W,C] NHWC, w [C,1,3,3]"""
    _chk(x, 'x'); _chk(w, 'w')
    n, h, wd, c = x.shape
    assert w.numel() == c * 9
    y = torch.empty((n, (h + stride - 1) // stride, (wd + stride - 1) // stride, c), dtype=torch.float32, device=x.device)
    check(_lib.lib().lp_dwconv3x3_fwd(x.data_ptr(), w.data_ptr(), _p(in_scale), _p(in_shift), y.data_ptr(), n, h, wd, c, stride, _stream()),
          'lp_dwconv3x3_fwd')
    return y


def pwconv(x: Tensor, w: Tensor, *, in_scale: Optional[Tensor] = None, in_shift: Optional[Tensor] = None, in_relu6: bool = False,
           in_res: Optional[Tensor] = None, want_x: bool = False, stats: bool = False):
    """fp32 1x1 conv y = a w^T on NHWC x, a = (relu6?)(x*in_scale+in_shift) (+ in_res).  -> (y, a | None, stats | None);
    ``stats`` = (partials, rows) for ``bn_finalize``"""
    _chk(x, 'x'); _chk(w, 'w')
    n, h, wd, k = x.shape
    cout = w.shape[0]
    assert w.numel() == cout * k
    pcount = n * h * wd
    y = torch.empty((n, h, wd, cout), dtype=torch.float32, device=x.device)
    xo = torch.empty_like(x) if want_x else None
    part, rows = None, 0
    if stats:
        rows = _lib.lib().lp_pwconv_stat_rows(pcount, k, cout)
        part = torch.empty(9 * rows * (cout // 4), dtype=torch.float32, device=x.device)
    if in_res is not None:
        _chk(in_res, 'in_res'); assert in_res.shape == x.shape
    check(_lib.lib().lp_pwconv_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), _p(in_scale), _p(in_shift), int(in_relu6), _p(in_res), _p(xo),
                                   _p(part), pcount, k, cout, _stream()), 'lp_pwconv_fwd')
    return y, xo, ((part, rows) if stats else None)


def dwconv3x3_stats(x: Tensor, w: Tensor, stride: int, in_scale: Optional[Tensor] = None, in_shift: Optional[Tensor] = None):
    """``dwconv3x3`` that also returns the BatchNorm partials of its output: (y, (partials, rows))"""
    _chk(x, 'x'); _chk(w, 'w')
    n, h, wd, c = x.shape
    rows = _lib.lib().lp_dwconv_stat_rows(n, h, wd, c, stride)
    assert rows > 0, c
    y = torch.empty((n, (h + stride - 1) // stride, (wd + stride - 1) // stride, c), dtype=torch.float32, device=x.device)
    part = torch.empty(9 * rows * (c // 4), dtype=torch.float32, device=x.device)
    check(_lib.lib().lp_dwconv3x3_stats_fwd(x.data_ptr(), w.data_ptr(), _p(in_scale), _p(in_shift), y.data_ptr(), part.data_ptr(), n, h, wd, c,
                                            stride, _stream()), 'lp_dwconv3x3_stats_fwd')
    return y, (part, rows)


def bn_finalize(stats, weight: Tensor, bias: Tensor, running_mean: Optional[Tensor], running_var: Optional[Tensor], momentum: float,
                eps: float) -> Tuple[Tensor, Tensor]:
    """train-mode BatchNorm (scale, shift) from the partials a conv launch left behind; updates the running statistics in place"""
    part, rows = stats
    c = weight.numel()
    out = torch.empty(2 * c, dtype=torch.float32, device=weight.device)
    check(_lib.lib().lp_bn_finalize(part.data_ptr(), rows, weight.data_ptr(), bias.data_ptr(), _p(running_mean), _p(running_var), out.data_ptr(),
                                    out[c:].data_ptr(), c, eps, momentum, _stream()), 'lp_bn_finalize')
    return out[:c], out[c:]


def affine_res(y: Tensor, scale: Tensor, shift: Tensor, res: Optional[Tensor] = None, prec: Optional[int] = None):
    """x = y*scale[c]+shift[c] (+res), NHWC; with ``prec`` also the operand planes of x -> (x, Act16)"""
    _chk(y, 'y')
    n, h, w, c = y.shape
    x = torch.empty_like(y)
    hi = lo = None
    if prec is not None:
        hi, lo = _alloc16(n, h, w, c, prec, y.device)
    check(_lib.lib().lp_affine_res(y.data_ptr(), scale.data_ptr(), shift.data_ptr(), _p(res), x.data_ptr(), _p(hi), _p(lo), n * h * w, c,
                                   prec if prec is not None else 0, _stream()), 'lp_affine_res')
    return x if prec is None else (x, Act16(hi, lo, c, None))


def affine_relu6_mean(y: Tensor, scale: Tensor, shift: Tensor) -> Tensor:
    _chk(y, 'y')
    n, h, w, c = y.shape
    out = torch.empty((n, c), dtype=torch.float32, device=y.device)
    check(_lib.lib().lp_affine_relu6_mean(y.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(), n, h * w, c, _stream()),
          'lp_affine_relu6_mean')
    return out


def bn_batch_affine(y: Tensor, weight: Tensor, bias: Tensor, running_mean: Optional[Tensor], running_var: Optional[Tensor],
                    momentum: float, eps: float) -> Tuple[Tensor, Tensor]:
    """train-mode BatchNorm2d of y [N,H,W,C] as a per-channel (scale, shift), updating the running statistics in place"""
    n, h, w, c = y.shape
    _chk(y, 'y')
    p = n * h * w
    scale = torch.empty(2 * c, dtype=torch.float32, device=y.device)
    ws = torch.empty(_lib.lib().lp_bn_stats_workspace_bytes(p, c) // 4, dtype=torch.float32, device=y.device)
    check(_lib.lib().lp_bn_stats(y.data_ptr(), weight.data_ptr(), bias.data_ptr(), _p(running_mean), _p(running_var), scale.data_ptr(),
                                 scale[c:].data_ptr(), ws.data_ptr(), p, c, eps, momentum, _stream()), 'lp_bn_stats')
    return scale[:c], scale[c:]


def sn_grad_apply(g: Tensor, w_orig: Tensor, u: Tensor, v: Tensor, sig: Tensor, accum: Optional[Tensor] = None) -> Optional[Tensor]:
    """gradient w.r.t. W_orig of a spectrally normalised layer from the raw gradient ``g`` w.r.t. W/sigma (legacy-hook autograd,
    u and v constants): g/sigma - <g, W_orig>/sigma^2 u v^T, in place on ``g`` -- or added to ``accum`` (returns None)."""
    _chk(g, 'g'); _chk(w_orig, 'w_orig')
    rows = g.shape[0]
    cols = g.numel() // rows
    dot = torch.empty(512, dtype=torch.float32, device=g.device)
    if accum is not None:
        assert accum.is_contiguous() and accum.dtype == torch.float32 and accum.numel() == g.numel()
    check(_lib.lib().lp_sn_grad_apply(g.data_ptr(), w_orig.data_ptr(), u.data_ptr(), v.data_ptr(), sig.data_ptr(), dot.data_ptr(), 0,
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
                   shift: Tensor, dgamma: Tensor, dbeta: Tensor, upsample: bool, amax: bool = False, planes: Optional[int] = None,
                   keep_dx: bool = True):
    """Backward of relu(AdaIN(x)) (+x2 upsample).  dgamma/dbeta: [N,C] views into the projector-output gradient (written).
    ``amax`` (here and below): the result will be packed as an fp16 gradient operand -- fold its max|.| into the kernel.
    ``planes`` = PREC_BF16 | PREC_BF16X3 (round 6): the result ALSO leaves as that mode's gradient operand planes, written by the same launch
    (lp_adain_relu_bwd_planes; no scale in those modes) -> (dx | None, Act16); ``keep_dx=False``: planes only."""
    _chk(dA, 'dA')
    if planes is not None and planes_direct_ok(planes, x, x.shape[3] if not isinstance(x, Act16) else 0):
        n, h, w, c = x.shape
        assert dA.shape == (n, h << int(upsample), w << int(upsample), c), (dA.shape, (n, h, w, c))
        assert gamma.stride(1) == 1 and dgamma.stride(1) == 1 and dbeta.stride(1) == 1 and gamma.stride(0) == dgamma.stride(0) == dbeta.stride(0)
        dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dA.device)          # (holds the masked gradient between the passes)
        ws = torch.empty(_lib.lib().lp_adain_bwd_workspace_bytes(n, h * w, c) // 4, dtype=torch.float32, device=dA.device)
        o_hi, o_lo = _alloc16(n, h, w, c, planes, dA.device)
        check(_lib.lib().lp_adain_relu_bwd_planes(dA.data_ptr(), x.data_ptr(), _p(add), gamma.data_ptr(), gamma.stride(0), mean.data_ptr(), rstd.data_ptr(),
                                                  scale.data_ptr(), shift.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(),
                                                  n, h, w, c, int(upsample), o_hi.data_ptr(), _p(o_lo), int(keep_dx), _stream()), 'lp_adain_relu_bwd_planes')
        return (dx if keep_dx else None), Act16(o_hi, o_lo, c, None)
    if planes is not None:          # (geometry the fused form does not cover: the two-launch form)
        dx = adain_relu_bwd(dA, x, add, gamma, mean, rstd, scale, shift, dgamma, dbeta, upsample, amax=amax)
        return dx, act_pack(dx, prec=planes, grad=True)
    x16 = x if isinstance(x, Act16) else None          # 16-bit-resident conv output (the generator's fp16 mode: no fp32 copy of x exists)
    if x16 is not None:
        assert x16.lo is None and x16.inv is None and x16.hi.shape[3] == x16.c, 'a 16-bit-resident x is ONE unscaled fp16 plane with C % 8 == 0'
        n, h, w, c = x16.hi.shape
    else:
        _chk(x, 'x')
        n, h, w, c = x.shape
    assert dA.shape == (n, h << int(upsample), w << int(upsample), c), (dA.shape, (n, h, w, c))
    assert gamma.stride(1) == 1 and dgamma.stride(1) == 1 and dbeta.stride(1) == 1
    assert gamma.stride(0) == dgamma.stride(0) == dbeta.stride(0)
    dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dA.device)
    ws = torch.empty(_lib.lib().lp_adain_bwd_workspace_bytes(n, h * w, c) // 4, dtype=torch.float32, device=dA.device)
    fn, xp, nm = (_lib.lib().lp_adain_relu_bwd, x.data_ptr(), 'lp_adain_relu_bwd') if x16 is None else \
        (_lib.lib().lp_adain_relu_bwd16, x16.hi.data_ptr(), 'lp_adain_relu_bwd16')
    check(fn(dA.data_ptr(), xp, _p(add), gamma.data_ptr(), gamma.stride(0), mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(),
             dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), n, h, w, c, int(upsample), _p(_amax_attach(dx, amax)), _stream()), nm)
    return dx


PLANES_DIRECT = os.environ.get('LP_PLANES_DIRECT', '1') != '0'      # 0: gradient operands of the bf16 / bf16x3 modes through lp_act_pack again (A/B knob)


def planes_direct_ok(prec: int, x, c: int) -> bool:
    return PLANES_DIRECT and prec in (PREC_BF16, PREC_BF16X3) and not isinstance(x, Act16) and c % 8 == 0


def sum2x2_planes(x: Tensor, prec: int) -> 'Act16':
    """sum2x2 straight to the bf16 / bf16x3 gradient operand planes (no fp32 result, no pack launch)"""
    _chk(x, 'x')
    n, h2, w2, c = x.shape
    if not planes_direct_ok(prec, x, c):
        return act_pack(sum2x2(x, amax=prec == PREC_F16), prec=prec, grad=True)
    o_hi, o_lo = _alloc16(n, h2 // 2, w2 // 2, c, prec, x.device)
    check(_lib.lib().lp_sum2x2_planes(x.data_ptr(), None, o_hi.data_ptr(), _p(o_lo), n, h2 // 2, w2 // 2, c, _stream()), 'lp_sum2x2_planes')
    return Act16(o_hi, o_lo, c, None)


# backward of nn.ConvPoolFn in one pass over the pooled gradient (round 6, lp_pool_grad_pack; 0: where + act_pack + avgpool2_bwd + act_pack)
POOL_GRAD_FUSED = os.environ.get('LP_POOL_GRAD_FUSED', '1') != '0'


def pool_grad_pack(dy: Tensor, y_mask: Optional[Tensor], prec: int, want_dm: bool, want_lo: bool, want_up: bool):
    """dm = dy * [y_mask > 0] -> (dm fp32 | None, operand planes of dm | None, operand planes of 0.25 * nearest_up2(dm) at twice the resolution | None),
    one launch (fp16 mode: + the amax partials of dy when its producer did not leave them)"""
    _chk(dy, 'dy')
    n, h, w, c = dy.shape
    assert c % 8 == 0 and (y_mask is None or tuple(y_mask.shape) == tuple(dy.shape))
    if y_mask is not None:
        _chk(y_mask, 'y_mask')
    dev = dy.device
    dm = torch.empty_like(dy) if want_dm else None
    lo_hi, lo_lo = _alloc16(n, h, w, c, prec, dev) if want_lo else (None, None)
    up_hi, up_lo = _alloc16(n, 2 * h, 2 * w, c, prec, dev) if want_up else (None, None)
    part = sc = None
    npart, pstride = 0, 1
    if prec == PREC_F16:
        rec = getattr(dy, '_lp_amax', None)
        if rec is not None and rec[1] == dy._version and FUSE_AMAX:
            part, npart, pstride = rec[0], _lib.lib().lp_amax_slots(), _lib.lib().lp_amax_slot_stride()
            sc = torch.empty(4, dtype=torch.float32, device=dev)
        else:
            npart = _lib.lib().lp_amax_blocks()
            buf = torch.empty(npart + 4, dtype=torch.float32, device=dev)
            part, sc = buf[:npart], buf[npart:]
            check(_lib.lib().lp_amax_partial(dy.data_ptr(), dy.numel(), part.data_ptr(), _stream()), 'lp_amax_partial')
    check(_lib.lib().lp_pool_grad_pack(dy.data_ptr(), _p(y_mask), _p(dm), _p(lo_hi), _p(lo_lo), _p(up_hi), _p(up_lo), n, h, w, c, prec,
                                       _p(part), npart, pstride, _p(sc), _stream()), 'lp_pool_grad_pack')
    lo16 = Act16(lo_hi, lo_lo, c, None if sc is None else sc[1:2]) if want_lo else None
    up16 = Act16(up_hi, up_lo, c, None if sc is None else sc[3:4]) if want_up else None
    return dm, lo16, up16


def sum2x2(x: Tensor, amax: bool = False) -> Tensor:
    _chk(x, 'x')
    n, h2, w2, c = x.shape
    out = torch.empty((n, h2 // 2, w2 // 2, c), dtype=torch.float32, device=x.device)
    check(_lib.lib().lp_sum2x2(x.data_ptr(), out.data_ptr(), n, h2 // 2, w2 // 2, c, _p(_amax_attach(out, amax)), _stream()), 'lp_sum2x2')
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


def head_bwd(t: Tensor, d_rgbs: Tensor, d_segm: Optional[Tensor], amax: bool = False) -> Tensor:
    _chk(t, 't'); _chk(d_rgbs, 'd_rgbs')
    n, h, w, _ = t.shape
    dz = torch.empty_like(t)
    if d_segm is not None:
        _chk(d_segm, 'd_segm')
    check(_lib.lib().lp_head_bwd(t.data_ptr(), d_rgbs.data_ptr(), _p(d_segm), dz.data_ptr(), n, h, w, _p(_amax_attach(dz, amax)), _stream()),
          'lp_head_bwd')
    return dz


def relu_bwd(dA: Tensor, x: Tensor) -> Tensor:
    _chk(dA, 'dA'); _chk(x, 'x')
    dx = torch.empty_like(x)
    check(_lib.lib().lp_relu_bwd(dA.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), _stream()), 'lp_relu_bwd')
    return dx


def avgpool2_fwd(x: Tensor, relu_in: bool, out16_prec: Optional[int] = None, relu_out: bool = False):
    """AvgPool2d(2) of relu?(x); ``relu_out``: y = relu(pool(x)).  ``out16_prec`` (PREC_F16 | PREC_BF16, C % 8 == 0): also the operand planes of y
    -> (y, Act16)"""
    _chk(x, 'x')
    n, h2, w2, c = x.shape
    y = torch.empty((n, h2 // 2, w2 // 2, c), dtype=torch.float32, device=x.device)
    fused = out16_prec is not None and out16_prec != PREC_BF16X3 and c % 8 == 0
    o_hi = torch.empty((n, h2 // 2, w2 // 2, c), dtype=torch.int16, device=x.device) if fused else None
    check(_lib.lib().lp_avgpool2_fwd(x.data_ptr(), y.data_ptr(), n, h2 // 2, w2 // 2, c, int(relu_in) | (2 if relu_out else 0), _p(o_hi),
                                     out16_prec if fused else 0, _stream()), 'lp_avgpool2_fwd')
    if out16_prec is None:
        return y
    return y, (Act16(o_hi, None, c, None) if fused else act_pack(y, pro=0, prec=out16_prec))


def avgpool2_fwd16(x: Act16, prec: int) -> Act16:
    """AvgPool2d(2) on operand planes (one-plane modes): planes in, planes out -- for no-grad chains that keep no fp32 activations"""
    n, h2, w2 = x.nhw
    c = x.c
    assert x.lo is None and c % 8 == 0 and x.hi.shape[3] == c and prec != PREC_BF16X3
    o_hi = torch.empty((n, h2 // 2, w2 // 2, c), dtype=torch.int16, device=x.hi.device)
    check(_lib.lib().lp_avgpool2_fwd16(x.hi.data_ptr(), o_hi.data_ptr(), n, h2 // 2, w2 // 2, c, prec, _stream()), 'lp_avgpool2_fwd16')
    return Act16(o_hi, None, c, None)


def avgpool2_bwd(dy: Tensor, x: Optional[Tensor], relu_in: bool, amax: bool = False, y_relu: Optional[Tensor] = None) -> Tensor:
    """backward of ``avgpool2_fwd``: ``x`` = the forward's input (read only with ``relu_in``); ``y_relu`` = the forward's OUTPUT when it applied
    ``relu_out`` (dy is masked by [y > 0]); with neither mask nothing of the forward needs to be kept"""
    _chk(dy, 'dy')
    assert not (relu_in and y_relu is not None)
    n, ho, wo, c = dy.shape
    h, w = 2 * ho, 2 * wo
    mask = None
    if relu_in:
        _chk(x, 'x'); assert tuple(x.shape) == (n, h, w, c)
        mask = x
    elif y_relu is not None:
        _chk(y_relu, 'y_relu'); assert tuple(y_relu.shape) == tuple(dy.shape)
        mask = y_relu
    dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dy.device)
    check(_lib.lib().lp_avgpool2_bwd(dy.data_ptr(), _p(mask), dx.data_ptr(), n, h, w, c, int(relu_in) | (2 if y_relu is not None else 0),
                                     _p(_amax_attach(dx, amax)), _stream()), 'lp_avgpool2_bwd')
    return dx


class Tap16(NamedTuple):
    """a feature tap kept as 16-bit operand planes (of relu(y): what the next conv consumes anyway) instead of an fp32 tensor"""
    act: Act16
    prec: int

    def float(self):
        """decoded fp32 NHWC tensor (debug tapes / tests)"""
        return self.act.hi.view(torch.float16 if self.prec == PREC_F16 else torch.bfloat16).float()

    def detach(self):
        return self


def l1_sum(a: Tensor, b, relu_in: bool, coef: float = 1.0, want_sign: bool = False):
    """coef * sum |relu?(a) - relu?(b)| as a 0-d tensor (block partials + a one-block finalize launch).  ``want_sign``: -> (term, sgn)
    with sgn int8 [numel] = the sign pattern the backward needs (``l1_bwd(sign=...)`` then does not read a and b again)"""
    _chk(a, 'a')
    buf = torch.empty(_lib.lib().lp_l1_partial_blocks() + 1, dtype=torch.float32, device=a.device)
    out = buf[-1:]
    sgn = torch.empty(a.numel(), dtype=torch.int8, device=a.device) if want_sign else None
    if isinstance(b, Tap16):           # b = 16-bit operand planes of the other image's tap (same NHWC element order, no channel padding)
        assert tuple(b.act.hi.shape) == tuple(a.shape) and b.act.lo is None, (b.act.hi.shape, a.shape)
        check(_lib.lib().lp_l1_fwd_b16(a.data_ptr(), b.act.hi.data_ptr(), b.prec, buf.data_ptr(), a.numel(), int(relu_in), float(coef),
                                       out.data_ptr(), _p(sgn), _stream()), 'lp_l1_fwd_b16')
        return (out.reshape(()), sgn) if want_sign else out.reshape(())
    _chk(b, 'b')
    assert a.shape == b.shape
    check(_lib.lib().lp_l1_fwd(a.data_ptr(), b.data_ptr(), buf.data_ptr(), a.numel(), int(relu_in), float(coef), out.data_ptr(), _p(sgn),
                               _stream()), 'lp_l1_fwd')
    return (out.reshape(()), sgn) if want_sign else out.reshape(())


def l1_sum16(a: 'Tap16', b: 'Tap16', coef: float = 1.0, want_sign: bool = False):
    """coef * sum |a - b| for two taps held as the operand planes of relu(.) (same precision mode); ``want_sign`` as ``l1_sum`` with relu_in"""
    assert a.prec == b.prec and tuple(a.act.hi.shape) == tuple(b.act.hi.shape) and a.act.lo is None and b.act.lo is None
    dev = a.act.hi.device
    buf = torch.empty(_lib.lib().lp_l1_partial_blocks() + 1, dtype=torch.float32, device=dev)
    out = buf[-1:]
    numel = a.act.hi.numel()
    sgn = torch.empty(numel, dtype=torch.int8, device=dev) if want_sign else None
    check(_lib.lib().lp_l1_fwd_ab16(a.act.hi.data_ptr(), b.act.hi.data_ptr(), a.prec, buf.data_ptr(), numel, float(coef), out.data_ptr(), _p(sgn),
                                    _stream()), 'lp_l1_fwd_ab16')
    return (out.reshape(()), sgn) if want_sign else out.reshape(())


def avgpool2_bwd_m16(dy: Tensor, mask: Act16, amax: bool = False) -> Tensor:
    """backward of AvgPool2d(2)(relu(x)) with the ReLU mask read from the operand planes of relu(x)"""
    _chk(dy, 'dy')
    n, h, w = mask.nhw
    c = mask.c
    assert mask.hi.shape[3] == c and tuple(dy.shape) == (n, h // 2, w // 2, c)
    dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dy.device)
    check(_lib.lib().lp_avgpool2_bwd_m16(dy.data_ptr(), mask.hi.data_ptr(), dx.data_ptr(), n, h, w, c, _p(_amax_attach(dx, amax)), _stream()),
          'lp_avgpool2_bwd_m16')
    return dx


def l1_bwd(a: Optional[Tensor], b: Optional[Tensor], grad_out: Tensor, coef: float, relu_in: bool, add: Optional[Tensor] = None,
           amax: bool = False, sign: Optional[Tensor] = None, shape=None) -> Tensor:
    """gradient of coef * sum|relu?(a) - relu?(b)| w.r.t. a, times grad_out; ``add`` (same shape) is summed in.  ``sign`` (from
    ``l1_sum(want_sign=True)``) replaces a and b; ``shape`` is then the shape of the result"""
    if sign is None:
        _chk(a, 'a'); _chk(b, 'b')
        shape = a.shape
    if add is not None:
        _chk(add, 'add'); assert tuple(add.shape) == tuple(shape)
    g = grad_out.reshape(1).contiguous().float()
    da = torch.empty(tuple(shape), dtype=torch.float32, device=g.device)
    check(_lib.lib().lp_l1_bwd(_p(a), _p(b), g.data_ptr(), float(coef), _p(add), da.data_ptr(), da.numel(), int(relu_in), _p(sign),
                               _p(_amax_attach(da, amax)), _stream()), 'lp_l1_bwd')
    return da



# ---- ResNeXt-50 identity encoder: BatchNorm statistics / backward, grouped 3x3 conv, stem, pooling (csrc/resnext.hip) -------------
def flat_hw(p: int) -> Tuple[int, int]:
    """(H, W) factorisation of P pixels for the 1x1 contractions (they do not care about the spatial structure): the widest W in
    {16, 8, 4} with H >= 2"""
    for w in (16, 8, 4):
        if p % w == 0 and p // w >= 2:
            return p // w, w
    raise ValueError(f'{p} pixels cannot be arranged as H x W with W in (16, 8, 4), H >= 2')


def flat16(a: Act16) -> Act16:
    """the same operand planes viewed as ONE image of P = N*H*W pixels (for 1x1 convs)"""
    c8 = a.hi.shape[-1]
    p = a.hi.numel() // c8
    h, w = flat_hw(p)
    return Act16(a.hi.view(1, h, w, c8), None if a.lo is None else a.lo.view(1, h, w, c8), a.c, a.inv)


def bn_train_stats(y: Tensor, gamma: Tensor, beta: Tensor, running_mean: Optional[Tensor], running_var: Optional[Tensor], momentum: float,
                   eps: float):
    """train-mode BatchNorm2d statistics of y [..., C] -> (mean, rstd, scale, shift) [C]; running statistics updated in place"""
    _chk(y, 'y')
    c = y.shape[-1]
    p = y.numel() // c
    out = torch.empty(4, c, dtype=torch.float32, device=y.device)
    ws = torch.empty(_lib.lib().lp_bn_train_stats_workspace_bytes(p, c) // 4, dtype=torch.float32, device=y.device)
    check(_lib.lib().lp_bn_train_stats(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, momentum, _p(running_mean), _p(running_var),
                                       out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), ws.data_ptr(), p, c, _stream()),
          'lp_bn_train_stats')
    return out[0], out[1], out[2], out[3]


def norm_act_bwd(dA: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, scale: Tensor, shift: Tensor, *, mask_mode: int = 0,
                 mask_src: Optional[Tensor] = None, want_g: bool = False, act_hi: float = 0.0, frozen: bool = False, amax: bool = False):
    """backward of act(BatchNorm(x)) over x [..., C] treated as [P][C] (per-channel statistics [C]):
    -> (dx, dgamma [C], dbeta [C], g | None).  mask_mode 0: act = ReLU (act_hi = 6: ReLU6) of the layer's own affine; 1: no activation;
    2: ReLU pattern taken from ``mask_src`` (the residual block's output).  ``want_g``: also return the masked incoming gradient."""
    _chk(dA, 'dA'); _chk(x, 'x')
    c = x.shape[-1]
    p = x.numel() // c
    assert dA.shape == x.shape, (dA.shape, x.shape)
    dx = torch.empty_like(x)
    dgb = torch.empty(2, c, dtype=torch.float32, device=x.device)
    g = torch.empty_like(x) if want_g else None
    ws = torch.empty(_lib.lib().lp_adain_bwd_workspace_bytes(1, p, c) // 4, dtype=torch.float32, device=x.device)
    if mask_src is not None:
        _chk(mask_src, 'mask_src'); assert mask_src.shape == x.shape
    check(_lib.lib().lp_norm_act_bwd(dA.data_ptr(), x.data_ptr(), None, gamma.data_ptr(), c, mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(),
                                     shift.data_ptr(), dx.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(), ws.data_ptr(), 1, p, 1, c, 0,
                                     mask_mode, _p(mask_src), _p(g), float(act_hi), int(frozen), _p(_amax_attach(dx, amax)), _stream()),
          'lp_norm_act_bwd')
    return dx, dgb[0], dgb[1], g


def pack_grouped(w: Tensor, mode: int, prec: int) -> WeightPack:
    """w [C, cg, 3, 3] (grouped conv, C/cg groups) -> the [9][CP][64] image of the block-diagonal formulation (lp_gconv16_fwd)"""
    _chk(w, 'w')
    c, cg = w.shape[0], w.shape[1]
    cp = _round_up(c, 128)
    hi = torch.empty((9, cp, 64), dtype=torch.int16, device=w.device)
    lo = torch.empty_like(hi) if prec == PREC_BF16X3 else None
    check(_lib.lib().lp_pack_grouped(w.data_ptr(), hi.data_ptr(), _p(lo), c, cg, cp, mode, _f16(prec), _stream()), 'lp_pack_grouped')
    return WeightPack(hi, lo, c, cg, cp, 64, 9)          # cols = the group size (logical contraction width per output channel)


def gconv16(a: Act16, pack: WeightPack, *, prec: int, amax: bool = False, stats: bool = False, want_y: bool = True, out16: bool = False):
    """grouped 3x3 conv (pad 1, stride 1) on operand planes a [N,H,W,C] -> y [N,H,W,C] fp32; with the mode-1 pack: the data gradient.
    ``stats``: -> (y, ConvStats | None) as ``conv16``.  ``out16``: the operand planes of y are a second output (y, Act16[, stats]);
    ``want_y=False`` (with ``out16``): no fp32 y is written, y is returned as None."""
    n, h, w = a.nhw
    c = a.c
    assert c == pack.rows and a.hi.shape[3] == c, (a.hi.shape, pack.rows)
    assert want_y or (out16 and not amax)
    dev = a.hi.device
    y = torch.empty((n, h, w, c), dtype=torch.float32, device=dev) if want_y else None
    o_hi = o_lo = None
    if out16:
        o_hi, o_lo = _alloc16(n, h, w, c, prec, dev)
    slots = _amax_attach(y, amax and prec == PREC_F16) if y is not None else None
    st_buf, st_rows, st_cap = None, None, 0
    if stats:
        import ctypes
        st_cap = _lib.lib().lp_conv16_stats_floats(n, h, w, c)
        st_buf = torch.empty(st_cap, dtype=torch.float32, device=dev)
        st_rows = ctypes.c_int(0)
    pl = 4 if prec == PREC_BF16X3 else 2
    nbytes = n * h * w * c * (pl + (4 if want_y else 0) + (pl if out16 else 0)) + pack.hi.numel() * pl
    with _Timed('gconv', 2.0 * n * h * w * c * pack.cols * 9, (n, h, w, pack.cols, c, 3, 0, 0), nbytes, mm=3 if prec == PREC_BF16X3 else 1):
        check(_lib.lib().lp_gconv16_fwd_planes(a.hi.data_ptr(), _p(a.lo), pack.hi.data_ptr(), _p(pack.lo), _p(y), _p(o_hi), _p(o_lo), _p(a.inv),
                                               n, h, w, c, pack.rows_p, pack.cols, prec, _p(slots), _p(st_buf), st_cap,
                                               None if st_rows is None else ctypes.addressof(st_rows), _stream()), 'lp_gconv16_fwd')
    out = (y,) if not out16 else (y, Act16(o_hi, o_lo, c, None))
    if stats:
        out = out + ((ConvStats(st_buf, st_rows.value, n) if st_rows.value > 0 else None),)
    return out[0] if len(out) == 1 else out


def bn_act16(y16: Act16, scale: Tensor, shift: Tensor, relu: bool = True) -> Act16:
    """operand planes of (relu?)(y*scale[c]+shift[c]) from a 16-BIT-RESIDENT conv output y16 (fp16 mode: the fp16 plane the conv epilogue
    wrote instead of fp32 y): ``act_pack`` pro 4 / 5 reading 2 B per element"""
    c = y16.c
    assert y16.lo is None and y16.inv is None and y16.hi.shape[-1] == c and c % 8 == 0
    hi = torch.empty_like(y16.hi)
    check(_lib.lib().lp_bn_act16(y16.hi.data_ptr(), scale.data_ptr(), shift.data_ptr(), hi.data_ptr(), y16.hi.numel() // c, c, int(relu),
                                 _stream()), 'lp_bn_act16')
    return Act16(hi, None, c, None)


def adain_act16(y16: Act16, scale: Tensor, shift: Tensor, relu: bool = True) -> Act16:
    """operand planes of (relu?)(y*scale[n,c]+shift[n,c]) from a 16-BIT-RESIDENT conv output y16 [N,H,W,C] (fp16 mode) with per-IMAGE affines
    [N, C] -- AdaIN + ReLU as the next conv's prologue (``act_pack`` pro 1 reading 2 B per element instead of 4)"""
    c = y16.c
    n = y16.hi.shape[0]
    assert y16.lo is None and y16.inv is None and y16.hi.shape[-1] == c and c % 8 == 0
    _chk(scale, 'scale'); _chk(shift, 'shift')
    assert tuple(scale.shape) == (n, c) and tuple(shift.shape) == (n, c), (scale.shape, shift.shape, (n, c))
    hi = torch.empty_like(y16.hi)
    check(_lib.lib().lp_adain_act16(y16.hi.data_ptr(), scale.data_ptr(), shift.data_ptr(), hi.data_ptr(), n, y16.hi.numel() // (n * c), c, int(relu),
                                    _stream()), 'lp_adain_act16')
    return Act16(hi, None, c, None)


def thin_wgrad16(x16: Act16, dy: Tensor, *, ksize: int, pro: int = 0, scale: Optional[Tensor] = None, shift: Optional[Tensor] = None,
                 splits: Optional[int] = None, sn=None, accum: Optional[Tensor] = None, bias_grad: bool = False):
    """``thin_wgrad`` of a conv with <= 4 OUTPUT channels whose wide input is a 16-bit-resident conv output (the generator head in the fp16 mode)"""
    _chk(dy, 'dy')
    n, h, w, cout = dy.shape
    cin = x16.c
    assert x16.lo is None and x16.inv is None and tuple(x16.hi.shape) == (n, h, w, cin)
    if splits is None:
        splits = default_splits(n, h, w, cin, cout)
    ws = torch.empty(_lib.lib().lp_conv_wgrad_workspace_bytes(cin, cout, ksize, splits) // 4, dtype=torch.float32, device=dy.device)
    dw = torch.empty((cout, cin, ksize, ksize), dtype=torch.float32, device=dy.device)
    with _Timed('wgrad_thin', 2.0 * n * h * w * cout * cin * ksize * ksize, (n, h, w, cin, cout, ksize, 0, pro)):
        check(_lib.lib().lp_thin_wgrad16(x16.hi.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), _p(scale), _p(shift), n, h, w, cin, cout,
                                         ksize, pro, splits, _stream()), 'lp_thin_wgrad16')
    dw = _sn_finish(dw, sn, accum)
    return (dw, dy.sum(dim=(0, 1, 2))) if bias_grad else dw


def y16_to_f32(y16: Act16) -> Tensor:
    """fp32 copy of a 16-bit-resident conv output (the rare consumers without a 16-bit form: statistics of a geometry the conv epilogue
    does not cover)"""
    return y16.hi.view(torch.float16)[..., :y16.c].float()


def norm_stats_finalize(st: ConvStats, n: int, c: int, gamma: Optional[Tensor], beta: Optional[Tensor], eps: float, *,
                        running_mean: Optional[Tensor] = None, running_var: Optional[Tensor] = None, momentum: float = 0.0):
    """(mean, rstd, scale, shift) [n, c] from the partials a conv epilogue wrote (``ConvStats``).  gamma/beta: [n, c] views (row stride
    free) or [c] (n == 1).  n == 1 with running statistics = train-mode BatchNorm."""
    dev = st.part.device
    out = torch.empty(4, n, c, dtype=torch.float32, device=dev)
    ab_stride = 0
    if gamma is not None:
        assert gamma.stride(-1) == 1 and beta.stride(-1) == 1
        ab_stride = gamma.stride(0) if gamma.dim() == 2 else c
        if beta.dim() == 2:
            assert beta.stride(0) == ab_stride
    assert n == st.images or n == 1, (n, st.images)
    rows = st.rows if n == st.images else st.rows * st.images       # n == 1 over several images: a BatchNorm over all pixels
    check(_lib.lib().lp_norm_stats_finalize(st.part.data_ptr(), rows, _p(gamma), _p(beta), ab_stride, eps, momentum, _p(running_mean),
                                            _p(running_var), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), n, c,
                                            _stream()), 'lp_norm_stats_finalize')
    if n == 1 and (gamma is None or gamma.dim() == 1):
        return out[0, 0], out[1, 0], out[2, 0], out[3, 0]
    return out[0], out[1], out[2], out[3]


def gconv_wgrad16(a: Act16, dy: Act16, group_size: int, *, prec: int, splits: Optional[int] = None) -> Tensor:
    """weight gradient [C, cg, 3, 3] of the grouped 3x3 conv (a = the planes the forward consumed, dy = packed output gradient)"""
    n, h, w = dy.nhw
    c = dy.c
    assert a.nhw == dy.nhw and a.c == c
    if splits is None:
        splits = max(1, min(512 // max(1, c // 64), (n * h * w + 127) // 128))
    ws = torch.empty(_lib.lib().lp_gconv_wgrad_workspace_bytes(c, splits) // 4, dtype=torch.float32, device=dy.hi.device)
    dw = torch.empty((c, group_size, 3, 3), dtype=torch.float32, device=dy.hi.device)
    pl = 4 if prec == PREC_BF16X3 else 2
    with _Timed('gconv_wgrad', 2.0 * n * h * w * c * group_size * 9, (n, h, w, group_size, c, 3, 0, 0), 2 * n * h * w * c * pl, mm=3 if prec == PREC_BF16X3 else 1):
        check(_lib.lib().lp_gconv16_wgrad(a.hi.data_ptr(), _p(a.lo), dy.hi.data_ptr(), _p(dy.lo), dw.data_ptr(), ws.data_ptr(), n, h, w, c,
                                          group_size, splits, prec, _p(dy.inv), _stream()), 'lp_gconv16_wgrad')
    return dw


def im2col_planes(x: Tensor, ksize: int, stride: int, pad: int, prec: int) -> Act16:
    """x [N,C,H,W] NCHW fp32 -> operand rows [N,Ho,Wo,K8] (K = C*k*k in nn.Conv2d weight order): a k x k conv becomes a 1x1 contraction"""
    _chk(x, 'x')
    n, c, h, w = x.shape
    ho, wo = (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1
    k = c * ksize * ksize
    hi, lo = _alloc16(n, ho, wo, k, prec, x.device)
    check(_lib.lib().lp_im2col_planes(x.data_ptr(), hi.data_ptr(), _p(lo), n, c, h, w, ksize, stride, pad, prec, _stream()), 'lp_im2col_planes')
    return Act16(hi, lo, k, None)


def bn_relu_maxpool(y: Tensor, scale: Tensor, shift: Tensor, prec: int, want_idx: bool = True):
    """MaxPool2d(3, 2, 1)(relu(y*scale[c]+shift[c])), y [N,H,W,C] -> (out fp32, its operand planes, argmax bytes | None)"""
    _chk(y, 'y')
    n, h, w, c = y.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = torch.empty((n, ho, wo, c), dtype=torch.float32, device=y.device)
    hi, lo = _alloc16(n, ho, wo, c, prec, y.device)
    idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=y.device) if want_idx else None
    check(_lib.lib().lp_bn_relu_maxpool_fwd(y.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(), hi.data_ptr(), _p(lo), _p(idx),
                                            n, h, w, c, prec, _stream()), 'lp_bn_relu_maxpool_fwd')
    return out, Act16(hi, lo, c, None), idx


def maxpool_bwd(dout: Tensor, idx: Tensor, h: int, w: int) -> Tensor:
    _chk(dout, 'dout')
    n, ho, wo, c = dout.shape
    dA = torch.empty((n, h, w, c), dtype=torch.float32, device=dout.device)
    check(_lib.lib().lp_maxpool_bwd(dout.data_ptr(), idx.data_ptr(), dA.data_ptr(), n, h, w, c, _stream()), 'lp_maxpool_bwd')
    return dA


def bn_add_act(y: Tensor, scale: Tensor, shift: Tensor, res: Optional[Tensor] = None, res_scale: Optional[Tensor] = None,
               res_shift: Optional[Tensor] = None, relu: bool = True, prec: Optional[int] = None, want_out: bool = True):
    """out = relu?(y*scale[c]+shift[c] + (res | res*res_scale[c]+res_shift[c])) over [..., C]; with ``prec`` -> (out, Act16).
    ``res`` as ``Act16`` (bf16 / bf16x3 planes of the same mode as ``prec``): the residual is read from the operand planes (lp_bn_add_act_planes);
    ``want_out=False`` then skips the fp32 copy -> (None, Act16)."""
    if isinstance(res, Act16):
        _chk(y, 'y')
        assert prec in (PREC_BF16, PREC_BF16X3) and res_scale is None and res.inv is None and (res.lo is not None) == (prec == PREC_BF16X3)
        assert tuple(res.hi.shape) == tuple(y.shape), (res.hi.shape, y.shape)
        c = y.shape[-1]
        out = torch.empty_like(y) if want_out else None
        hi = torch.empty(y.shape, dtype=torch.int16, device=y.device)
        lo = torch.empty_like(hi) if prec == PREC_BF16X3 else None
        check(_lib.lib().lp_bn_add_act_planes(y.data_ptr(), scale.data_ptr(), shift.data_ptr(), res.hi.data_ptr(), _p(res.lo), _p(out), hi.data_ptr(), _p(lo),
                                              y.numel() // c, c, int(relu), prec, _stream()), 'lp_bn_add_act_planes')
        return out, Act16(hi, lo, c, None)
    y16 = y if isinstance(y, Act16) else None
    if y16 is not None:          # 16-bit-resident conv output (fp16 mode)
        assert prec in (None, PREC_F16) and y16.lo is None and y16.inv is None
        shape, dev = y16.hi.shape, y16.hi.device
    else:
        _chk(y, 'y')
        shape, dev = y.shape, y.device
    c = shape[-1]
    p = 1
    for d in shape[:-1]:
        p *= d
    assert want_out or prec is not None
    out = torch.empty(shape, dtype=torch.float32, device=dev) if want_out else None
    hi = lo = None
    if prec is not None:
        hi = torch.empty(shape, dtype=torch.int16, device=dev)
        lo = torch.empty_like(hi) if prec == PREC_BF16X3 else None
    if res is not None:
        _chk(res, 'res'); assert res.shape == shape, (res.shape, shape)
    if y16 is not None:
        check(_lib.lib().lp_bn_add_act16(y16.hi.data_ptr(), scale.data_ptr(), shift.data_ptr(), _p(res), _p(res_scale), _p(res_shift),
                                         _p(out), _p(hi), p, c, int(relu), _stream()), 'lp_bn_add_act16')
    else:
        check(_lib.lib().lp_bn_add_act(y.data_ptr(), scale.data_ptr(), shift.data_ptr(), _p(res), _p(res_scale), _p(res_shift), _p(out),
                                       _p(hi), _p(lo), p, c, int(relu), prec if prec is not None else 0, _stream()), 'lp_bn_add_act')
    return out if prec is None else (out, Act16(hi, lo, c, None))


def subsample2(x: Tensor) -> Tensor:
    """x [N,H,W,C] -> x[:, ::2, ::2] contiguous (fp32 or int16 planes with C*itemsize % 16 == 0)"""
    n, h, w, c = x.shape
    assert x.is_contiguous()
    out = torch.empty((n, (h + 1) // 2, (w + 1) // 2, c), dtype=x.dtype, device=x.device)
    check(_lib.lib().lp_subsample2(x.data_ptr(), out.data_ptr(), n, h, w, c * x.element_size(), _stream()), 'lp_subsample2')
    return out


def subsample2_16(a: Act16) -> Act16:
    return Act16(subsample2(a.hi), None if a.lo is None else subsample2(a.lo), a.c, a.inv)


def zero_stuff2_16(a: Act16, h: int, w: int) -> Act16:
    """adjoint of ``subsample2`` on operand planes: [N,ceil(h/2),ceil(w/2),C8] -> [N,h,w,C8], zeros at the odd positions"""
    n, hs, ws_, c8 = a.hi.shape
    assert hs == (h + 1) // 2 and ws_ == (w + 1) // 2

    def one(t):
        out = torch.empty((n, h, w, c8), dtype=t.dtype, device=t.device)
        check(_lib.lib().lp_zero_stuff2(t.data_ptr(), out.data_ptr(), n, h, w, c8 * t.element_size(), _stream()), 'lp_zero_stuff2')
        return out
    return Act16(one(a.hi), None if a.lo is None else one(a.lo), a.c, a.inv)


def add_strided2(d: Tensor, s: Tensor) -> Tensor:
    """d[:, ::2, ::2] += s, in place (fp32 NHWC)"""
    _chk(d, 'd'); _chk(s, 's')
    n, h, w, c = d.shape
    assert s.shape == (n, (h + 1) // 2, (w + 1) // 2, c)
    check(_lib.lib().lp_add_strided2(d.data_ptr(), s.data_ptr(), n, h, w, c, _stream()), 'lp_add_strided2')
    return d


def proj_score_fwd(out: Tensor, embed: Optional[Tensor]):
    """critic head: out [N,H,W,C] -> (pooled = sum_hw relu(out) [N,C], dot = <pooled, embed> [N] | None), one launch"""
    _chk(out, 'out')
    n, h, w, c = out.shape
    if embed is not None:
        _chk(embed, 'embed'); assert tuple(embed.shape) == (n, c)
    pooled = torch.empty((n, c), dtype=torch.float32, device=out.device)
    dot = torch.empty((n,), dtype=torch.float32, device=out.device) if embed is not None else None
    check(_lib.lib().lp_proj_score_fwd(out.data_ptr(), _p(embed), pooled.data_ptr(), _p(dot), n, h * w, c, _stream()), 'lp_proj_score_fwd')
    return pooled, dot


def proj_score_bwd(out: Tensor, embed: Optional[Tensor], pooled: Tensor, g_pooled: Optional[Tensor], g_dot: Optional[Tensor], want_out: bool, want_embed: bool):
    n, h, w, c = out.shape
    for t, nm in ((g_pooled, 'g_pooled'), (g_dot, 'g_dot')):
        if t is not None:
            _chk(t, nm)
    d_out = torch.empty_like(out) if want_out else None
    d_embed = torch.empty_like(pooled) if (want_embed and g_dot is not None) else None
    check(_lib.lib().lp_proj_score_bwd(out.data_ptr(), _p(embed), pooled.data_ptr(), _p(g_pooled), _p(g_dot if embed is not None else None), _p(d_out),
                                       _p(d_embed), n, h * w, c, _stream()), 'lp_proj_score_bwd')
    return d_out, d_embed


def image_prep_fwd(x: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """x NCHW [N,3,H,W] in [-1, 1] -> ((x + 1) / 2 - mean) / std as NHWC [N,H,W,3] (one launch)"""
    _chk(x, 'x'); _chk(mean, 'mean'); _chk(std, 'std')
    n, c, h, w = x.shape
    assert c == 3 and mean.numel() == 3 and std.numel() == 3
    out = torch.empty((n, h, w, 3), dtype=torch.float32, device=x.device)
    check(_lib.lib().lp_image_prep_fwd(x.data_ptr(), mean.data_ptr(), std.data_ptr(), out.data_ptr(), n, h * w, _stream()), 'lp_image_prep_fwd')
    return out


def image_prep_bwd(g: Tensor, std: Tensor) -> Tensor:
    _chk(g, 'g')
    n, h, w, c = g.shape
    assert c == 3
    dx = torch.empty((n, 3, h, w), dtype=torch.float32, device=g.device)
    check(_lib.lib().lp_image_prep_bwd(g.data_ptr(), std.data_ptr(), dx.data_ptr(), n, h * w, _stream()), 'lp_image_prep_bwd')
    return dx


def spatial_mean(x: Tensor) -> Tensor:
    _chk(x, 'x')
    n, h, w, c = x.shape
    out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    check(_lib.lib().lp_spatial_mean_fwd(x.data_ptr(), out.data_ptr(), n, h * w, c, _stream()), 'lp_spatial_mean_fwd')
    return out


def spatial_mean_bwd(g: Tensor, h: int, w: int) -> Tensor:
    _chk(g, 'g')
    n, c = g.shape
    dx = torch.empty((n, h, w, c), dtype=torch.float32, device=g.device)
    check(_lib.lib().lp_spatial_mean_bwd(g.data_ptr(), dx.data_ptr(), n, h * w, c, _stream()), 'lp_spatial_mean_bwd')
    return dx


# ---- MobileNetV2 backward: depthwise 3x3 ------------------------------------------------------------------------------------------
def dwconv3x3_dgrad(dy: Tensor, w: Tensor, h: int, wd: int, stride: int) -> Tensor:
    """gradient w.r.t. the activated input of ``dwconv3x3`` ([N,h,wd,C]) from dy [N,ceil(h/s),ceil(wd/s),C]"""
    _chk(dy, 'dy'); _chk(w, 'w')
    n, ho, wo, c = dy.shape
    assert ho == (h + stride - 1) // stride and wo == (wd + stride - 1) // stride and w.numel() == c * 9
    da = torch.empty((n, h, wd, c), dtype=torch.float32, device=dy.device)
    check(_lib.lib().lp_dwconv3x3_dgrad(dy.data_ptr(), w.data_ptr(), da.data_ptr(), n, h, wd, c, stride, _stream()), 'lp_dwconv3x3_dgrad')
    return da


def dwconv3x3_wgrad(x: Tensor, dy: Tensor, stride: int, in_scale: Optional[Tensor] = None, in_shift: Optional[Tensor] = None) -> Tensor:
    """weight gradient [C,1,3,3] of ``dwconv3x3`` (x = its raw input, activation relu6(x*in_scale+in_shift) recomputed on load)"""
    _chk(x, 'x'); _chk(dy, 'dy')
    n, h, wd, c = x.shape
    dw = torch.empty((c, 1, 3, 3), dtype=torch.float32, device=x.device)
    ws = torch.empty(_lib.lib().lp_dwconv3x3_wgrad_workspace_bytes(c) // 4, dtype=torch.float32, device=x.device)
    check(_lib.lib().lp_dwconv3x3_wgrad(x.data_ptr(), _p(in_scale), _p(in_shift), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), n, h, wd, c,
                                        stride, _stream()), 'lp_dwconv3x3_wgrad')
    return dw


def bn_bwd16(dA: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, scale: Tensor, shift: Tensor, *, prec: int, mask_mode: int = 0,
             mask_src: Optional[Tensor] = None, want_g: bool = False, act_hi: float = 0.0, frozen: bool = False):
    """backward of act(BatchNorm(x)) over x [..., C] written straight to the OPERAND PLANES of dy (what the weight / data gradient
    contractions consume): -> (Act16 of dy, dgamma [C], dbeta [C], g | None).  No fp32 dy, no masked-gradient temporary; in fp16 mode the
    planes carry the power-of-two scale of lp_bn_bwd16 (``Act16.inv``).  Arguments as ``norm_act_bwd``; ``x`` may be a 16-bit-resident
    conv output (``Act16``, fp16 mode) and ``mask_src`` (mask_mode 2) the operand planes of the activated tensor instead of its fp32 copy."""
    _chk(dA, 'dA')
    x16 = x if isinstance(x, Act16) else None          # 16-bit-resident conv output (fp16 mode)
    if x16 is not None:
        assert prec == PREC_F16 and x16.lo is None and x16.inv is None
        shape = x16.hi.shape
    else:
        _chk(x, 'x')
        shape = x.shape
    c = shape[-1]
    p = dA.numel() // c
    assert dA.shape == shape and c % 8 == 0, (dA.shape, shape)
    dev = dA.device
    hi = torch.empty(shape, dtype=torch.int16, device=dev)
    lo = torch.empty_like(hi) if prec == PREC_BF16X3 else None
    small = torch.empty(2 + 2 * c, dtype=torch.float32, device=dev)
    sc, dg, db = small[:2], small[2:2 + c], small[2 + c:]
    g = torch.empty(shape, dtype=torch.float32, device=dev) if want_g else None
    ws = torch.empty(_lib.lib().lp_bn_bwd16_workspace_bytes(p, c) // 4, dtype=torch.float32, device=dev)
    mptr = None
    if isinstance(mask_src, Act16):                    # the ReLU pattern from the planes of the activated tensor (2 B per element)
        assert mask_mode == 2 and mask_src.hi.shape == shape, (mask_mode, mask_src.hi.shape, shape)
        mask_mode, mptr = 3, mask_src.hi.data_ptr()
    elif mask_src is not None:
        _chk(mask_src, 'mask_src'); assert mask_src.shape == shape
        mptr = mask_src.data_ptr()
    check(_lib.lib().lp_bn_bwd16_h(dA.data_ptr(), None if x16 is not None else x.data_ptr(), None if x16 is None else x16.hi.data_ptr(), mptr,
                                   gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), hi.data_ptr(), _p(lo),
                                   sc.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), p, c, mask_mode, float(act_hi), int(frozen), prec,
                                   _p(g), _stream()), 'lp_bn_bwd16')
    return Act16(hi, lo, c, sc[1:] if prec == PREC_F16 else None), dg, db, g
