import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_generator_module import load, make_gen, rel
from oracle import lp_oracle as O
z = load('generator_small.npz')
# oracle with intermediate capture
sd = {k[3:]: torch.from_numpy(v).clone() for k, v in z.items() if k.startswith('sd.')}
for k, v in sd.items():
    if k.endswith('weight_orig') or k.endswith('.bias') or k.endswith('.constant'): v.requires_grad_(True)
inter = []
orig_rb = O.resblock_ada
def rb(x, *a, **k):
    x.retain_grad(); inter.append(x)
    return orig_rb(x, *a, **k)
O.resblock_ada = rb
orig_adain = O.adain
e = torch.from_numpy(z['embeds']).requires_grad_(True); p = torch.from_numpy(z['pose']).requires_grad_(True)
image_size, nc, mx, _, _ = (int(v) for v in z['cfg'])
# capture head input: last block output = input of final adain; hook via wrapping adain calls count
calls = []
def ad(x, g, b, eps=1e-4):
    if x.requires_grad and not x.is_leaf: x.retain_grad()
    calls.append(x)
    return orig_adain(x, g, b, eps)
O.adain = ad
rgb, segm = O.generator_forward(sd, e, p, num_channels=nc, max_num_channels=mx, image_size=image_size, train=True)
((rgb * torch.from_numpy(z['r1'])).sum() + (segm * torch.from_numpy(z['r2'])).sum()).backward()
print('oracle adain calls', len(calls), 'blocks', len(inter))
G = make_gen(z, prec=1)
G.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('sd.')}, strict=True)
G = G.cuda().train(); G._debug = {}
ec = torch.from_numpy(z['embeds']).cuda().requires_grad_(True); pc = torch.from_numpy(z['pose']).cuda().requires_grad_(True)
dd = dict(embeds=ec, pose_embedding=pc); G(dd)
((dd['fake_rgbs'] * torch.from_numpy(z['r1']).cuda()).sum() + (dd['fake_segm'] * torch.from_numpy(z['r2']).cuda()).sum()).backward()
d = G._debug
nb = len(inter)
# adain call order: per block norm0(x), norm1(h1); then head
print('dx_head (grad wrt last block out):', rel(d[f'dx{nb}'].permute(0,3,1,2), calls[2*nb].grad))
for bi in range(nb-1, -1, -1):
    print(f'block {bi}: dh1 {rel(d[f"dh1_{bi}"].permute(0,3,1,2), calls[2*bi+1].grad):.3e}  dx {rel(d[f"dx{bi}"].permute(0,3,1,2), inter[bi].grad):.3e}')
# ---- deeper: head stage
print('---- head stage')
import torch.nn.functional as F
convs = []
real_conv = F.conv2d
def cv(x, w, b=None, *a, **k):
    if x.requires_grad and not x.is_leaf: x.retain_grad()
    convs.append(x); return real_conv(x, w, b, *a, **k)
O.F.conv2d = cv
sd2 = {k[3:]: torch.from_numpy(v).clone() for k, v in z.items() if k.startswith('sd.')}
for k, v in sd2.items():
    if k.endswith('weight_orig') or k.endswith('.bias') or k.endswith('.constant'): v.requires_grad_(True)
e2 = torch.from_numpy(z['embeds']).requires_grad_(True); p2 = torch.from_numpy(z['pose']).requires_grad_(True)
calls.clear()
rgb, segm = O.generator_forward(sd2, e2, p2, num_channels=nc, max_num_channels=mx, image_size=image_size, train=True)
((rgb * torch.from_numpy(z['r1'])).sum() + (segm * torch.from_numpy(z['r2'])).sum()).backward()
a_head = convs[-1]
print('dA_head vs oracle grad of activated head input:', rel(d['dA_head'].permute(0,3,1,2), a_head.grad))
xh = calls[-1]
print('x_head (last block out) match:', rel(G._debug_x.permute(0,3,1,2), xh) if hasattr(G,'_debug_x') else 'n/a')
# recompute adain bwd in torch from the oracle's a_head.grad and xh
x = xh.detach().double(); dA = a_head.grad.double()
mean = x.mean((2,3),keepdim=True); var = x.var((2,3),unbiased=False,keepdim=True); r = 1/torch.sqrt(var+1e-4)
print('oracle dx_head norm', xh.grad.norm().item(), ' mine', d[f'dx{nb}'].norm().item())
dxm = d[f'dx{nb}'].permute(0,3,1,2).cpu().double(); dxo = xh.grad.double()
diff = (dxm - dxo)
print('diff per-sample-channel rel:', (diff.flatten(2).norm(dim=2) / dxo.flatten(2).norm(dim=2)))
print('diff mean over hw / |dx| :', diff.mean((2,3)) / dxo.flatten(2).norm(dim=2)*32)
