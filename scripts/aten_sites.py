"""Which Python lines still launch stock ATen kernels inside one eager training step?  A TorchDispatchMode records, for every ATen call that
touches a GPU tensor, the operator, the argument shapes and the innermost frame of this package on the Python stack (autograd-engine calls of
built-in backward nodes have none: they are listed by operator and shape).  View / metadata operators are skipped.
usage (GPU box): WORKLOAD=metatrain|finetune python scripts/aten_sites.py > gpurun_out/aten_sites.txt"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench

SKIP = ('view', 'reshape', 'permute', 'transpose', 'expand', 'slice', 'select', 'squeeze', 'unsqueeze', 'detach', 'alias', 'as_strided', 't.default', 'empty', 'unbind',
        'split', 'narrow', '_unsafe_view', 'size', 'stride', 'is_', 'record_stream', 'lift_fresh', 'new_empty', '_local_scalar_dense', 'item', 'chunk', 'unflatten', 'flatten.using_ints')


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.hist = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(s in name for s in SKIP):
            return out
        flat = [a for a in torch.utils._pytree.tree_leaves((args, kwargs)) if torch.is_tensor(a)]
        if not any(t.is_cuda for t in flat):
            return out
        site = '(no package frame: autograd engine)'
        for fr in reversed(traceback.extract_stack()[:-1]):
            if 'latent_pose_reenactment_amd' in fr.filename or fr.filename.endswith('bench.py'):
                site = f'{fr.filename.split("latent_pose_reenactment_amd/")[-1]}:{fr.lineno} {fr.line[:70]}'
                break
        shapes = ','.join('x'.join(map(str, t.shape)) or '()' for t in flat[:3])
        self.hist[(name.replace('aten.', ''), site, shapes)] += 1
        return out


torch.cuda.set_device(0)
args = bench.make_args(256, 8, 'cuda:0', 1, 0, os.environ.get('LP_PREC', 'f16'), finetune=os.environ.get('WORKLOAD', 'metatrain') == 'finetune')
tm, opt_G, opt_D, holycow = bench.build(args)
data, target = bench.synthetic_batch(args, 8, seed=123)
for _ in range(2):
    holycow.train_step(tm, data, target, opt_G, opt_D, args)
torch.cuda.synchronize()
with Log() as log:
    holycow.train_step(tm, data, target, opt_G, opt_D, args)
torch.cuda.synchronize()
total = sum(log.hist.values())
print(f'{total} ATen calls on GPU tensors in one step (views excluded)')
by_site = collections.defaultdict(list)
for (name, site, shapes), n in log.hist.items():
    by_site[site].append((n, name, shapes))
for site, items in sorted(by_site.items(), key=lambda kv: -sum(i[0] for i in kv[1])):
    print(f'{sum(i[0] for i in items):4d}  {site}')
    for n, name, shapes in sorted(items, reverse=True)[:12]:
        print(f'        {n:3d} x {name:32s} {shapes}')
