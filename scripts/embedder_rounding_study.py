"""CPU study behind the gates of tests/test_resnext_hip.py: the ResNeXt HIP path's orchestration (embedders/resnext_hip.py) run through the
plain-torch emulation of its kernels (tests/emu_ops.py) in fp64, with the operands of EVERY contraction rounded the way the kernels round
them (bf16x3: hi + lo bf16;  f16: fp16, gradient operands scaled by the power of two that puts their amax into [2^12, 2^13)) -- i.e. the
error the precision mode itself implies on this network and input, independent of any kernel.  usage: embedder_rounding_study.py bf16x3|f16 [noy16]
Measured (shallow [2,1,1,1] net, 8 structured 128 x 128 frames; logits / all gradients vs exact fp64):
  bf16x3: eval 6.7e-6 / 3.0e-4, train-mode BatchNorm 3.6e-5 / 1.5e-2;   f16 noy16: eval 4.9e-4 / 1.7e-3, train 3.0e-3 / 1.3e-1;
  f16 (conv outputs rounded to fp16 as the 16-bit-resident path stores them): eval 5.0e-4 / 1.7e-3, train 3.2e-3 / 1.4e-1
The GPU kernels land on the same figures (profiles/README.md), so the residual is conditioning, not kernel error."""
import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/latent_pose_reenactment_amd'); sys.path.insert(0,'/root/repo/tests')
import torch, copy
import emu_ops
from embedders import resnext_hip
from embedders.backbones import ResNeXt
from latent_pose_reenactment_amd import hipops
resnext_hip.ops = emu_ops
hipops.PackBatch = emu_ops.PackBatch; hipops.pack_grouped = emu_ops.pack_grouped
MODE = sys.argv[1]
import os
os.environ['LP_PREC_E'] = MODE
if len(sys.argv) > 2 and sys.argv[2] == 'noy16':      # fp16 mode with fp32-resident conv outputs (LP_E_Y16=0)
    resnext_hip.Y16 = False
def rnd(t):
    if MODE == 'f16': return t.to(torch.float16).to(t.dtype)
    if MODE == 'bf16x3':
        hi = t.to(torch.bfloat16).to(t.dtype); lo = (t - hi).to(torch.bfloat16).to(t.dtype); return hi + lo
    return t
# wrap contraction ops with operand rounding (weights rounded too)
def wrapA(a): return emu_ops.Act16(rnd(a.hi), None, a.c, a.inv)
def wrapP(p): return emu_ops.Pack(rnd(p.w), p.mode)
o_conv16, o_wg, o_g, o_gw = emu_ops.conv16, emu_ops.conv_wgrad16, emu_ops.gconv16, emu_ops.gconv_wgrad16
def rnd_out(r):          # 16-bit-resident conv outputs: the fp16 plane the epilogue writes
    if isinstance(r, tuple):
        return tuple(emu_ops.Act16(rnd(t.hi), None, t.c, t.inv) if isinstance(t, emu_ops.Act16) else t for t in r)
    return r
emu_ops.conv16 = lambda a, pack, **kw: rnd_out(o_conv16(wrapA(a), wrapP(pack), **kw))
emu_ops.conv_wgrad16 = lambda a, dy, **kw: o_wg(wrapA(a), wrapA(dy), **kw)
emu_ops.gconv16 = lambda a, pack, **kw: rnd_out(o_g(wrapA(a), wrapP(pack), **kw))
emu_ops.gconv_wgrad16 = lambda a, dy, cg, **kw: o_gw(wrapA(a), wrapA(dy), cg, **kw)
# note: f16 gradient operands are amax-scaled in the real path: emulate with scaling to [2^12,2^13)
if MODE == 'f16':
    o_ap = emu_ops.act_pack
    def ap(x, *, pro=0, scale=None, shift=None, prec=0, grad=False):
        r = o_ap(x, pro=pro, scale=scale, shift=shift, prec=prec, grad=grad)
        if grad:
            m = r.hi.abs().max(); import math
            e = math.frexp(float(m))[1]; s = 2.0 ** (13 - e)
            return emu_ops.Act16((r.hi * s).to(torch.float16).to(r.hi.dtype) / s, None, r.c, None)
        return r
    emu_ops.act_pack = ap
torch.manual_seed(7)
for train in (False, True):
    m = ResNeXt([2,1,1,1], 32, 4, 32).double()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.uniform_(-0.3, 0.3)
            mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 1.5)
    m2 = copy.deepcopy(m); m.train(train); m2.train(train)
    from test_resnext_hip import structured_frames
    x = structured_frames(8, 128, 3).double()
    r = torch.randn(8, 32, dtype=torch.double)
    from oracle import backbones_ref as BR
    y_ref = BR.resnext_forward(m, x); (y_ref * r).sum().backward()
    m2._hip_structure()
    y = resnext_hip.ResNeXtFunction.apply(m2, x, *[p for _, p in m2.named_parameters()])
    (y * r).sum().backward()
    def rel(a, b): return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    num = sum((p2.grad - p1.grad).norm()**2 for p1, p2 in zip(m.parameters(), m2.parameters())); den = sum(p1.grad.norm()**2 for p1 in m.parameters())
    print(MODE, 'train' if train else 'eval', 'logits', f'{rel(y, y_ref):.2e}', 'all-grads', f'{(num/den).sqrt().item():.2e}')
    worst = sorted(((rel(p2.grad, p1.grad), k) for (k, p1), (_, p2) in zip(m.named_parameters(), m2.named_parameters())), reverse=True)[:6]
    print('  worst', [(k, f'{e:.1e}') for e, k in worst])
