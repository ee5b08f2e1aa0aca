"""CPU emulation of the MFMA operand precisions on the full-size generator (oracle arithmetic, operands rounded before every
conv in forward AND backward).  Decides which operand format meets the 1e-3 rel-L2 gate with ONE MFMA per MAC.
usage: python scripts/precision_study.py [B]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import lp_oracle as O

MODE = ["fp32"]
_orig = F.conv2d

def rnd(t, scaled=False):
    m = MODE[0]
    if m == 'fp32':
        return t
    if m == 'bf16':
        return t.to(torch.bfloat16).float()
    if m == 'f16':
        if scaled:           # gradient operand: power-of-two scale from the tensor's amax (target amax 2^12)
            amax = t.abs().max().item()
            if amax == 0: return t
            s = 2.0 ** (12 - math.ceil(math.log2(amax)))
            return (t * s).to(torch.float16).float() / s
        return t.to(torch.float16).float()
    if m == 'f16ns':         # fp16 without gradient scaling
        return t.to(torch.float16).float()
    raise ValueError(m)

class RConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad):
        m = MODE[0]
        if m in ('A', 'C'):      # forward f16
            MODE[0] = 'f16'; xr, wr = rnd(x), rnd(w); MODE[0] = m
        elif m == 'B':
            xr, wr = x, w
        else:
            xr, wr = rnd(x), rnd(w)
        ctx.save_for_backward(xr, wr); ctx.sp = (stride, pad, b is not None)
        return _orig(xr, wr, b, stride, pad)
    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors; stride, pad, hb = ctx.sp
        m = MODE[0]
        if m == 'A' or m == 'C':
            dyr = dy              # A: exact backward; C: dy split hi+lo (exact), other operand f16 (xr, wr already rounded)
        elif m == 'B':
            MODE[0] = 'f16'; dyr = rnd(dy, scaled=True); xr, wr = rnd(xr), rnd(wr); MODE[0] = m
        else:
            dyr = rnd(dy, scaled=True)
        dx = torch.nn.grad.conv2d_input(xr.shape, wr, dyr, stride, pad)
        dw = torch.nn.grad.conv2d_weight(xr, wr.shape, dyr, stride, pad)
        return dx, dw, (dy.sum((0, 2, 3)) if hb else None), None, None


def conv2d(x, w, b=None, stride=1, padding=0, *a, **k):
    return RConv.apply(x, w, b, stride, padding)

MASKS = {'rec': None, 'use': None, 'i': 0}
_relu = torch.relu
def relu_hook(x):
    if MASKS['use'] is not None:
        m = MASKS['use'][MASKS['i']]; MASKS['i'] += 1
        return x * m
    y = _relu(x)
    if MASKS['rec'] is not None:
        MASKS['rec'].append((x > 0).float())
    return y

def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()

LOSS = ['proj']
def run(sd0, e, p, r1, r2, mode, S):
    MODE[0] = mode
    sd = {k: v.clone() for k, v in sd0.items()}
    for k, v in sd.items():
        if k.endswith('weight_orig') or k.endswith('.bias') or k.endswith('.constant'):
            v.requires_grad_(True)
    eo, po = e.clone().requires_grad_(True), p.clone().requires_grad_(True)
    O.F.conv2d = conv2d
    torch.relu = relu_hook; MASKS['i'] = 0
    try:
        rgb, segm = O.generator_forward(sd, eo, po, num_channels=64, max_num_channels=512, image_size=S, train=False)
        if LOSS[0] == 'proj':
            ((rgb * r1).sum() + (segm * r2).sum()).backward()
        elif LOSS[0] == 'ones':
            (rgb.sum() + segm.sum()).backward()
        else:
            t = F.avg_pool2d(F.pad(r1, (8, 8, 8, 8), mode='reflect'), 17, 1) * 3 + 0.5
            sgm = (F.avg_pool2d(F.pad(r2, (8, 8, 8, 8), mode='reflect'), 17, 1) > 0).float()
            ((rgb - t * sgm).abs().mean() + (segm - sgm).abs().mean()).backward()
    finally:
        O.F.conv2d = _orig; torch.relu = _relu
    grads = {k: v.grad for k, v in sd.items() if v.grad is not None}
    grads['d_embeds'] = eo.grad; grads['d_pose'] = po.grad
    return rgb.detach(), segm.detach(), grads

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sys.path.insert(0, '/root/repo')
    # random-init state dict with the reference key names: borrow the shapes from the product module definition (CPU construction only)
    from latent_pose_reenactment_amd.nn import Generator
    G = Generator('zero', 3, 4, 64, 512, 512, 256, 'in', 4, 2, S, prec=0)
    with torch.no_grad():
        G.constant.constant.normal_()
    sd0 = {k: v.detach().clone() for k, v in G.state_dict().items()}
    e, p = torch.randn(B, 512), torch.randn(B, 256)
    # settle the power iteration
    with torch.no_grad():
        for _ in range(5):
            O.generator_forward(sd0, e, p, num_channels=64, max_num_channels=512, image_size=S, train=True)
    g = torch.Generator().manual_seed(1)
    r1, r2 = torch.randn(B, 3, S, S, generator=g), torch.randn(B, 1, S, S, generator=g)
    for LOSS[0] in ['proj']:
      print('loss =', LOSS[0])
      ref = run(sd0, e, p, r1, r2, 'fp32', S)
      for mode in ['f16', 'A', 'B', 'C']:
        MASKS['rec'] = []; MASKS['use'] = None
        out = run(sd0, e, p, r1, r2, mode, S)
        MASKS['use'] = MASKS['rec']; MASKS['rec'] = None
        refm = run(sd0, e, p, r1, r2, 'fp32', S)
        MASKS['use'] = None
        gm = {k: rel(out[2][k], refm[2][k]) for k in ref[2] if not k.endswith('skip.1.bias')}
        print(f'   tie-masked: rgb {rel(out[0], refm[0]):.3e} grads median {sorted(gm.values())[len(gm)//2]:.3e} worst {sorted(gm.items(), key=lambda kv: -kv[1])[:3]}')
        ge = {k: rel(out[2][k], ref[2][k]) for k in ref[2] if not k.endswith('skip.1.bias')}
        worst = sorted(ge.items(), key=lambda kv: -kv[1])[:5]
        print(f'{mode}: rgb {rel(out[0], ref[0]):.3e} segm {rel(out[1], ref[1]):.3e} | grads: median {sorted(ge.values())[len(ge)//2]:.3e} worst {[(k, f"{v:.2e}") for k, v in worst]}', flush=True)

main()
