"""List the ATen operators (with Python call sites) that still launch stock PyTorch kernels inside one eager training step
(WORKLOAD=metatrain (default) | finetune; LP_PREC as bench.py: default f16).
usage (GPU box): python scripts/aten_ops.py > gpurun_out/aten_ops.txt"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

torch.cuda.set_device(0)
args = bench.make_args(256, 8, 'cuda:0', 1, 0, os.environ.get('LP_PREC', 'f16'), finetune=os.environ.get('WORKLOAD', 'metatrain') == 'finetune')
tm, opt_G, opt_D, holycow = bench.build(args)
data, target = bench.synthetic_batch(args, 8, seed=123)
for _ in range(3):
    holycow.train_step(tm, data, target, opt_G, opt_D, args)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    holycow.train_step(tm, data, target, opt_G, opt_D, args)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.device_time_total <= 0 or ev.cpu_children and any(c.name.startswith('aten::') and c.device_time_total > 0 for c in ev.cpu_children):
        continue
    site = next((s for s in (ev.stack or []) if 'latent_pose_reenactment_amd' in s or 'bench.py' in s), '(autograd engine / other)')
    site = site.split('latent_pose_reenactment_amd/')[-1]
    a = agg[ev.name]
    a[0] += 1; a[1] += ev.device_time_total; a[2][site[:110]] += 1
print(f'{"op":34s} {"calls":>6s} {"gpu_us":>9s}   top call sites')
for name, (n, us, sites) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f'{name:34s} {n:6d} {us:9.0f}   ' + ' | '.join(f'{s} x{c}' for s, c in sites.most_common(8)))
