"""where does a new host batch per step cost time?  resident | host memcpy only | H2D of pre-pinned buffers only | full prefetcher"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))
import torch
import bench
from latent_pose_reenactment_amd.dataloaders.prefetch import DevicePrefetcher
args = bench.make_args(256, 8, 'cuda:0', 1, 0, 'f16', finetune=True)
args.generator = 'vector_pose_unsupervised_segmentation_noBottleneck'
tm, opt_G, opt_D, holycow = bench.build(args)
data, target = bench.synthetic_batch(args, 8, seed=123)
step = holycow.GraphedTrainStep(tm, opt_G, opt_D, args, data, target, warmup_steps=3)
host = [({k: v.cpu().clone() for k, v in data.items()}, {k: v.cpu().clone() for k, v in target.items()}) for i in range(4)]
pinned = [tuple({k: v.pin_memory() for k, v in d.items()} for d in b) for b in host]
N = 30
def timed(f):
    f(3); torch.cuda.synchronize(); t0 = time.perf_counter(); f(N); torch.cuda.synchronize(); return (time.perf_counter() - t0) / N * 1e3
def resident(n):
    for _ in range(n): step()
def memcpy_only(n):
    bufs = {k: torch.empty_like(v).pin_memory() for k, v in host[0][0].items()}
    for i in range(n):
        for k, v in host[i % 4][0].items(): bufs[k].copy_(v)
        step()
side = torch.cuda.Stream()
dev = [tuple({k: torch.empty_like(v, device='cuda') for k, v in d.items()} for d in b) for b in host[:2]]
def h2d_only(n):
    for i in range(n):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for dd, pd in zip(dev[i % 2], pinned[i % 4]):
                for k in pd: dd[k].copy_(pd[k], non_blocking=True)
        torch.cuda.current_stream().wait_stream(side) if False else None
        step()
def h2d_same_stream(n):
    for i in range(n):
        for dd, pd in zip(dev[i % 2], pinned[i % 4]):
            for k in pd: dd[k].copy_(pd[k], non_blocking=True)
        step()
def full(n):
    for d, t in DevicePrefetcher((host[i % 4] for i in range(n)), 'cuda:0'):
        step.load_batch(d, t); step()
def full_pinned_source(n):
    for d, t in DevicePrefetcher((pinned[i % 4] for i in range(n)), 'cuda:0'):
        step.load_batch(d, t); step()
def load_batch_only(n):
    for i in range(n):
        step.load_batch(*dev[i % 2]); step()
def h2d_side_nowait(n):
    for i in range(n):
        with torch.cuda.stream(side):
            for dd, pd in zip(dev[i % 2], pinned[i % 4]):
                for k in pd: dd[k].copy_(pd[k], non_blocking=True)
        step()
ev_lb = [None]
def h2d_side_after_launch(n):
    # the copy of batch i+1 is issued AFTER step i has been enqueued and only waits for the event recorded behind step i-1's load_batch
    for i in range(n):
        step.load_batch(*dev[i % 2])
        e = torch.cuda.Event(); e.record()
        step()
        with torch.cuda.stream(side):
            if ev_lb[0] is not None: side.wait_event(ev_lb[0])
            for dd, pd in zip(dev[(i + 1) % 2], pinned[(i + 1) % 4]):
                for k in pd: dd[k].copy_(pd[k], non_blocking=True)
        ev_lb[0] = e
flat_pin = torch.empty(sum(v.numel() * v.element_size() for d in host[0] for v in d.values()) + 64, dtype=torch.uint8).pin_memory()
flat_dev = [torch.empty_like(flat_pin, device='cuda') for _ in range(2)]
def h2d_one_flat_copy_side(n):
    for i in range(n):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            flat_dev[i % 2].copy_(flat_pin, non_blocking=True)
        step()
def h2d_one_flat_copy_same(n):
    for i in range(n):
        flat_dev[i % 2].copy_(flat_pin, non_blocking=True)
        step()
for name, f in (('resident', resident), ('side stream, no stream coupling at all', h2d_side_nowait), ('side stream, issued after the step launch', h2d_side_after_launch),
                ('one flat 25 MB H2D on a side stream', h2d_one_flat_copy_side), ('one flat 25 MB H2D on the compute stream', h2d_one_flat_copy_same), ('host memcpy to pinned only', memcpy_only), ('async H2D on a side stream only', h2d_only), ('H2D on the compute stream', h2d_same_stream),
                ('device-to-device load_batch only', load_batch_only), ('full prefetcher', full), ('full prefetcher, pinned source', full_pinned_source), ('resident again', resident)):
    print(f'{name:40s} {timed(f):8.3f} ms/step', flush=True)
mb = sum(v.numel() * v.element_size() for d in host[0] for v in d.values()) / 1e6
print(f'batch = {mb:.1f} MB')
