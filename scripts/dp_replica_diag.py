"""Which parameters differ between data-parallel replicas, and after which step?  Two (or more) ranks run the real meta-training step with
parallel.GradReducer; after every step the per-parameter bit checksums are gathered and the names of the parameters whose replicas differ
are printed (none expected: same start by broadcast, same averaged gradients, same optimizer arithmetic).
usage (one GPU shared by gloo ranks): python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 \
           scripts/dp_replica_diag.py [eager|graph] [steps=3] [image_size=128] [backend=gloo]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'eager'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
size = int(sys.argv[3]) if len(sys.argv) > 3 else 128
backend = sys.argv[4] if len(sys.argv) > 4 else 'gloo'
world, rank = int(os.environ['WORLD_SIZE']), int(os.environ['RANK'])
dev = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
torch.cuda.set_device(dev)
dist.init_process_group(backend=backend, init_method='env://')
args = bench.make_args(size, 8, f'cuda:{dev}', world, rank, os.environ.get('LP_PREC', 'f16'), finetune=False)
tm, opt_G, opt_D, holycow = bench.build(args)
from latent_pose_reenactment_amd.parallel import GradReducer  # noqa: E402
tm.reducer = GradReducer(tm, finetune=False, optimizer_G=opt_G, optimizer_D=opt_D, max_batch=8)
data, target = bench.synthetic_batch(args, 8, seed=123 + rank)
named = [(f'{m}.{k}', p) for m in ('generator', 'embedder', 'discriminator') for k, p in getattr(tm, m).named_parameters()]
# plus the spectral-norm power-iteration vectors of the critic (buffers: rank-local by design, but identical as long as the weights are)
named += [(f'discriminator.{k} (buffer)', b) for k, b in tm.discriminator.named_buffers() if b.dtype == torch.float32]


def checksums():
    out = []
    for _, p in named:
        b = p.detach().reshape(-1).view(torch.int32).to(torch.int64)
        out.append(int(b.sum().item()) ^ int((b * (torch.arange(b.numel(), device=b.device) % 8191 + 1)).sum().item()))
    return out


def compare(tag):
    mine = checksums()
    allc = [None] * world
    dist.all_gather_object(allc, mine)
    if rank == 0:
        bad = [named[i][0] for i in range(len(named)) if any(c[i] != allc[0][i] for c in allc)]
        print(f'[replicas] {tag}: {len(bad)} of {len(named)} tensors differ between ranks' + (': ' + ', '.join(bad[:12]) + (' ...' if len(bad) > 12 else '') if bad else ''), flush=True)
    # how far apart: max |delta| of the label embedding between rank 0 and the others (0 expected)
    w = tm.discriminator.embed.weight_orig.detach()
    ref = w.clone()
    dist.broadcast(ref, 0)
    d = (w - ref).abs()
    stat = torch.stack([d.max(), (d > 0).sum().float(), (d > 0).any(dim=1).sum().float()])
    dist.all_reduce(stat, op=dist.ReduceOp.MAX)
    if rank == 0 and float(stat[0]) > 0:
        print(f'[replicas]    label embedding vs rank 0: max |delta| {float(stat[0]):.3e}, {int(stat[1])} elements in {int(stat[2])} rows differ', flush=True)


compare('after the start-up broadcast')
if mode == 'graph':
    step = holycow.GraphedTrainStep(tm, opt_G, opt_D, args, data, target, warmup_steps=1)
    compare('after 1 eager warm-up step + capture')
else:
    def step():
        holycow.train_step(tm, data, target, opt_G, opt_D, args)
for i in range(steps):
    step()
    torch.cuda.synchronize()
    compare(f'after {mode} step {i + 1}')
if mode == 'graph' and os.environ.get('DIAG_EAGER_AFTER', '1') != '0':
    # what bench.py does after its timed replays: two EAGER one-stream steps (the instrumented steps of its live roofline)
    keep = os.environ.get('LP_OVERLAP')
    os.environ['LP_OVERLAP'] = '0'
    for i in range(2):
        holycow.train_step(tm, data, target, opt_G, opt_D, args)
        torch.cuda.synchronize()
        compare(f'after eager one-stream step {i + 1} following the replays')
    if keep is None:
        os.environ.pop('LP_OVERLAP', None)
    else:
        os.environ['LP_OVERLAP'] = keep
dist.barrier()
dist.destroy_process_group()
