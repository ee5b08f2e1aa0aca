"""Identity encoder (ResNeXt-50 32x4d, train-mode BatchNorm) at the FULL configs[2] geometry -- 8 samples x 8 frames of 256 x 256, the
initialisation bench.py uses -- against the stock layers in fp64 on the same device: per-frame logits, `embeds` (mean over the 8 frames,
what north_star's 1e-3 tolerance is about), all parameter gradients.  One precision mode per process (LP_PREC_E is read at import).
usage: LP_PREC_E=f16|bf16x3 python scripts/e1_parity_full.py [frames=64] [size=256]"""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))
import torch  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
content = sys.argv[3] if len(sys.argv) > 3 else 'noise'       # noise: U[0,1) frames (what bench.py feeds) | smooth: low-frequency content + 10 % noise
from embedders import backbones  # noqa: E402
from oracle import backbones_ref as BR  # noqa: E402
from dataloaders.synthetic_voxceleb2 import make_sample  # noqa: E402

torch.manual_seed(123)
net = backbones.resnext50_32x4d(num_classes=512).cuda().train()
ref = copy.deepcopy(net).double()
k = 8
b = frames // k
datas = [make_sample(i, size, k, 98000, False, 123)[0] for i in range(b)]
x = torch.stack([d['enc_rgbs'] for d in datas]).cuda().reshape(b * k, 3, size, size)
if content == 'smooth':
    g = torch.Generator().manual_seed(3)
    low = torch.rand(b * k, 3, 8, 8, generator=g)
    x = (torch.nn.functional.interpolate(low, size=(size, size), mode='bilinear', align_corners=False)
         + 0.1 * torch.rand(b * k, 3, size, size, generator=g)).clamp(0, 1).cuda()
r = torch.randn(b, 512, device='cuda')


def rel(a, c):
    a, c = a.double(), c.double()
    return ((a - c).norm() / c.norm().clamp_min(1e-30)).item()


t0 = time.time()
y = net(x)
assert net.__dict__.get('_hip_param_names') is not None, 'the HIP path did not run'
emb = y.view(b, k, -1).mean(1)
(emb * r).sum().backward()
torch.cuda.synchronize()
yr = BR.resnext_forward(ref, x.double())
embr = yr.view(b, k, -1).mean(1)
(embr * r.double()).sum().backward()
torch.cuda.synchronize()
# calibration: the stock fp32 layers (MIOpen / rocBLAS -- the arithmetic class of the reference itself) against the same fp64 run
m32 = copy.deepcopy(ref).float()
for p_ in m32.parameters():
    p_.grad = None
y32 = BR.resnext_forward(m32, x)
e32 = y32.view(b, k, -1).mean(1)
(e32 * r).sum().backward()
n32 = sum((p.grad.double() - q.grad).norm() ** 2 for p, q in zip(m32.parameters(), ref.parameters()))
num = sum((p.grad.double() - q.grad).norm() ** 2 for p, q in zip(net.parameters(), ref.parameters()))
den = sum(q.grad.norm() ** 2 for q in ref.parameters())
n32 = n32
ga = torch.cat([p.grad.double().reshape(-1) for p in net.parameters()]); gb = torch.cat([q.grad.reshape(-1) for q in ref.parameters()])
for p_ in net.parameters():
    p_.grad = None


def step():
    yy = net(x)
    (yy.view(b, k, -1).mean(1) * r).sum().backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
modes = [{0: 'bf16', 1: 'bf16x3', 2: 'f16'}[m] for m in net.block_precs()]
print(f'[e1-parity] encoder base mode {modes[0]}, fp16 tail {sum(m == "f16" for m in modes) if modes[0] != "f16" else 16} blocks, head fp16 kinds [{os.environ.get("LP_E_HEAD_F16", "")}], Y16={os.environ.get("LP_E_Y16", "1")}, eager fwd+bwd {ms:.2f} ms: '
      f'{frames} frames {size}px train-mode BN: per-frame logits {rel(y, yr):.3e}  embeds {rel(emb, embr):.3e}  '
      f'all-gradients {float((num / den).sqrt()):.3e} (cosine {float((ga * gb).sum() / (ga.norm() * gb.norm())):.4f})  | stock fp32 layers vs fp64: logits {rel(y32, yr):.3e} embeds {rel(e32, embr):.3e} all-gradients {float((n32 / den).sqrt()):.3e}  [{content}; {time.time() - t0:.0f} s]', flush=True)
