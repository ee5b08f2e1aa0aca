"""Plain-torch (CPU or GPU, fp32/fp64) emulation of the ``latent_pose_reenactment_amd.hipops`` entry points that the embedder's HIP
path (embedders/resnext_hip.py, embedders/mobilenet_hip.py) is composed of.  TEST INFRASTRUCTURE ONLY:
  * on CPU it lets the orchestration (which kernel, on which tensor, in which order; what is saved for backward) be checked against
    the stock nn.Module autograd without a GPU: tests monkeypatch ``resnext_hip.ops`` with this module;
  * on the GPU the same functions are the per-op references of tests/test_resnext_hip.py.
"Operand planes" are ordinary float tensors here (``Act16.hi`` = the activated tensor itself), weight packs carry the weight and the
orientation."""
from typing import NamedTuple, Optional

import torch
import torch.nn.functional as F


class Act16(NamedTuple):
    hi: torch.Tensor
    lo: Optional[torch.Tensor]
    c: int
    inv: Optional[torch.Tensor]

    @property
    def nhw(self):
        return tuple(self.hi.shape[:3])


class Pack(NamedTuple):
    w: torch.Tensor
    mode: int


def flat_hw(p):
    for w in (16, 8, 4):
        if p % w == 0 and p // w >= 2:
            return p // w, w
    raise ValueError(p)


def flat16(a):
    c = a.hi.shape[-1]
    h, w = flat_hw(a.hi.numel() // c)
    return Act16(a.hi.reshape(1, h, w, c), None, a.c, a.inv)


class PackBatch:
    def __init__(self, specs, prec):
        self.prec = prec
        self.key = tuple((w.data_ptr(), m, bool(sk)) for w, m, sk in specs)
        self.specs = specs

    def update(self):
        return [Pack(w, m) for w, m, _ in self.specs]


def pack_grouped(w, mode, prec):
    return Pack(w, mode)


def act_pack(x, *, pro=0, scale=None, shift=None, prec=0, grad=False):
    if pro == 0:
        v = x
    elif pro == 2:
        v = torch.relu(x)
    elif pro in (3, 4, 5):
        v = x * scale + shift
        v = torch.clamp(v, 0, 6) if pro == 3 else (torch.relu(v) if pro == 4 else v)
    else:
        raise ValueError(pro)
    return Act16(v, None, x.shape[-1], None)


class ConvStats(NamedTuple):
    part: torch.Tensor          # (emulation: the conv output itself)
    rows: int


def conv16(a, pack, *, ksize, upsample=False, bias=None, res=None, res_shift=0, alpha=None, prec=0, relu_mask=None, out16=None, amax=False,
           stats=False, want_y=True, kind=None):
    assert ksize == 1 and not upsample
    w = pack.w.reshape(pack.w.shape[0], -1)                       # [Cout, Cin]
    x = a.hi[..., :a.c]
    y = x @ (w.t() if pack.mode == 0 else w)
    if bias is not None:
        y = y + bias
    if res is not None:
        y = y + res
    out = (y if want_y else None,)
    if out16 is not None:
        out = out + (Act16(torch.relu(y) if out16 else y, None, y.shape[-1], None),)
    if stats:
        out = out + (ConvStats(y, 1),)
    return out[0] if len(out) == 1 else out


def norm_stats_finalize(st, n, c, gamma, beta, eps, *, running_mean=None, running_var=None, momentum=0.0):
    assert n == 1
    return bn_train_stats(st.part, gamma, beta, running_mean, running_var, momentum, eps)


def conv_wgrad16(a, dy, *, ksize, upsample=False, prec=0, splits=None, sn=None, accum=None, bias_grad=False, bias_accum=None, kind=None):
    assert ksize == 1
    x = a.hi[..., :a.c].reshape(-1, a.c)
    d = dy.hi[..., :dy.c].reshape(-1, dy.c)
    dw = (d.t() @ x).reshape(dy.c, a.c, 1, 1)
    return (dw, d.sum(0)) if bias_grad else dw


def gconv16(a, pack, *, prec=0, amax=False, stats=False, want_y=True, out16=False):
    w = pack.w
    groups = a.c // w.shape[1]
    x = a.hi.permute(0, 3, 1, 2)
    y = F.conv2d(x, w, padding=1, groups=groups) if pack.mode == 0 else F.conv_transpose2d(x, w, padding=1, groups=groups)
    y = y.permute(0, 2, 3, 1).contiguous()
    out = (y if want_y else None,)
    if out16:
        out = out + (Act16(y, None, y.shape[-1], None),)
    if stats:
        out = out + (ConvStats(y, 1),)
    return out[0] if len(out) == 1 else out


def bn_act16(y16, scale, shift, relu=True):
    v = y16.hi * scale + shift
    return Act16(torch.relu(v) if relu else v, None, y16.c, None)


def y16_to_f32(y16):
    return y16.hi


def gconv_wgrad16(a, dy, group_size, *, prec=0, splits=None):
    c = a.c
    x, d = a.hi.permute(0, 3, 1, 2), dy.hi.permute(0, 3, 1, 2)
    return torch.nn.grad.conv2d_weight(x, (c, group_size, 3, 3), d, padding=1, groups=c // group_size)


def bn_train_stats(y, gamma, beta, running_mean, running_var, momentum, eps):
    c = y.shape[-1]
    y2 = y.reshape(-1, c)
    mean = y2.mean(0)
    var = y2.var(0, unbiased=False)
    if running_mean is not None:
        n = y2.shape[0]
        running_mean.mul_(1 - momentum).add_(momentum * mean)
        running_var.mul_(1 - momentum).add_(momentum * var * (n / max(n - 1, 1)))
    rstd = (var + eps).rsqrt()
    scale = gamma * rstd
    return mean, rstd, scale, beta - mean * scale


def norm_act_bwd(dA, x, gamma, mean, rstd, scale, shift, *, mask_mode=0, mask_src=None, want_g=False, act_hi=0.0, frozen=False, amax=False):
    c = x.shape[-1]
    if mask_mode == 0:
        a = x * scale + shift
        m = a > 0
        if act_hi > 0:
            m = m & (a < act_hi)
        g = dA * m
    elif mask_mode == 1:
        g = dA
    else:
        g = dA * (mask_src > 0)
    xhat = (x - mean) * rstd
    g2, xh2 = g.reshape(-1, c), xhat.reshape(-1, c)
    dbeta = g2.sum(0)
    dgamma = (g2 * xh2).sum(0)
    if frozen:
        dx = g * (gamma * rstd)
    else:
        p = g2.shape[0]
        dx = (gamma * rstd) * (g - dbeta / p - xhat * (dgamma / p))
    return dx, dgamma, dbeta, (g if want_g else None)


def im2col_planes(x, ksize, stride, pad, prec):
    n, c, h, w = x.shape
    ho, wo = (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1
    cols = F.unfold(x, ksize, padding=pad, stride=stride)          # [N, C*k*k, Ho*Wo], k index = (c, ky, kx)
    return Act16(cols.transpose(1, 2).reshape(n, ho, wo, c * ksize * ksize).contiguous(), None, c * ksize * ksize, None)


def bn_relu_maxpool(y, scale, shift, prec, want_idx=True):
    a = torch.relu(y * scale + shift).permute(0, 3, 1, 2)
    out, idx = F.max_pool2d(a, 3, 2, 1, return_indices=True)
    out = out.permute(0, 2, 3, 1).contiguous()
    return out, Act16(out, None, out.shape[-1], None), idx


def maxpool_bwd(dout, idx, h, w):
    d = dout.permute(0, 3, 1, 2)
    n, c = d.shape[:2]
    dA = torch.zeros(n, c, h * w, dtype=d.dtype, device=d.device)
    dA.scatter_add_(2, idx.reshape(n, c, -1), d.reshape(n, c, -1))
    return dA.view(n, c, h, w).permute(0, 2, 3, 1).contiguous()


def bn_add_act(y, scale, shift, res=None, res_scale=None, res_shift=None, relu=True, prec=None, want_out=True):
    if isinstance(y, Act16):
        y = y.hi
    if isinstance(res, Act16):          # lp_bn_add_act_planes: the residual from the operand planes of the block input, fp32 out optional
        assert res_scale is None and prec is not None
        v = y * scale + shift + res.hi
        if relu:
            v = torch.relu(v)
        return (v if want_out else None), Act16(v, None, v.shape[-1], None)
    v = y * scale + shift
    if res is not None:
        v = v + (res * res_scale + res_shift if res_scale is not None else res)
    if relu:
        v = torch.relu(v)
    return v if prec is None else ((v if want_out else None), Act16(v, None, v.shape[-1], None))


def subsample2(x):
    return x[:, ::2, ::2].contiguous()


def subsample2_16(a):
    return Act16(subsample2(a.hi), None, a.c, a.inv)


def zero_stuff2_16(a, h, w):
    n, hs, ws, c = a.hi.shape
    out = torch.zeros(n, h, w, c, dtype=a.hi.dtype, device=a.hi.device)
    out[:, ::2, ::2] = a.hi
    return Act16(out, None, a.c, a.inv)


def add_strided2(d, s):
    d[:, ::2, ::2] += s
    return d


def spatial_mean(x):
    return x.mean(dim=(1, 2))


def spatial_mean_bwd(g, h, w):
    n, c = g.shape
    return (g / (h * w))[:, None, None, :].expand(n, h, w, c).contiguous()


# ---- MobileNetV2 pieces ---------------------------------------------------------------------------------------------------------
def dwconv3x3(x, w, stride, in_scale=None, in_shift=None):
    a = x if in_scale is None else torch.clamp(x * in_scale + in_shift, 0, 6)
    c = x.shape[-1]
    y = F.conv2d(a.permute(0, 3, 1, 2), w.reshape(c, 1, 3, 3), stride=stride, padding=1, groups=c)
    return y.permute(0, 2, 3, 1).contiguous()


def dwconv3x3_dgrad(dy, w, h, wd, stride):
    c = dy.shape[-1]
    x = torch.zeros(dy.shape[0], c, h, wd, dtype=dy.dtype, device=dy.device)
    da = torch.nn.grad.conv2d_input(x.shape, w.reshape(c, 1, 3, 3), dy.permute(0, 3, 1, 2), stride=stride, padding=1, groups=c)
    return da.permute(0, 2, 3, 1).contiguous()


def dwconv3x3_wgrad(x, dy, stride, in_scale=None, in_shift=None):
    a = x if in_scale is None else torch.clamp(x * in_scale + in_shift, 0, 6)
    c = x.shape[-1]
    return torch.nn.grad.conv2d_weight(a.permute(0, 3, 1, 2), (c, 1, 3, 3), dy.permute(0, 3, 1, 2), stride=stride, padding=1, groups=c)


def affine_relu6_mean(y, scale, shift):
    return torch.clamp(y * scale + shift, 0, 6).mean(dim=(1, 2))


def bn_bwd16(dA, x, gamma, mean, rstd, scale, shift, *, prec=0, mask_mode=0, mask_src=None, want_g=False, act_hi=0.0, frozen=False):
    if isinstance(x, Act16):
        x = x.hi
    if isinstance(mask_src, Act16):
        mask_src = mask_src.hi
    dx, dg, db, g = norm_act_bwd(dA, x, gamma, mean, rstd, scale, shift, mask_mode=mask_mode, mask_src=mask_src, want_g=want_g, act_hi=act_hi,
                                 frozen=frozen)
    return Act16(dx, None, x.shape[-1], None), dg, db, g
