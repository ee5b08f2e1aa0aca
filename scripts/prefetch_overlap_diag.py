"""input path beside the multi-stream captured step: resident batch vs a new host batch per step through DevicePrefetcher.
usage: prefetch_overlap_diag.py finetune|metatrain   (knobs: LP_OVERLAP_ENCODERS, LP_OVERLAP_CRITERIONS, LP_COPY_PRIORITY, GPU_MAX_HW_QUEUES)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))
import torch
import bench
from latent_pose_reenactment_amd.dataloaders.prefetch import DevicePrefetcher
wl = sys.argv[1]
args = bench.make_args(256, 8, 'cuda:0', 1, 0, 'f16', finetune=(wl == 'finetune'))
if wl == 'finetune':
    args.generator = 'vector_pose_unsupervised_segmentation_noBottleneck'
tm, opt_G, opt_D, holycow = bench.build(args)
data, target = bench.synthetic_batch(args, 8, seed=123)
step = holycow.GraphedTrainStep(tm, opt_G, opt_D, args, data, target, warmup_steps=3)
host = [({k: v.cpu().clone() for k, v in data.items()}, {k: v.cpu().clone() for k, v in target.items()}) for i in range(4)]
N = 30
def timed(f):
    f(5); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); f(N); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / N * 1e3)
    return best
def resident(n):
    for _ in range(n): step()
for _ in range(int(os.environ.get('SKIP_STREAMS', '0'))):       # shift which pool stream (-> hardware queue) the copy stream gets
    torch.cuda.Stream()
last = [None]
def full(n):
    pf = DevicePrefetcher((host[i % 4] for i in range(n)), 'cuda:0')
    for d, t in pf:
        step.load_batch(d, t); step()
    last[0] = {k: round(v / n * 1e3, 3) for k, v in pf.waits.items()}
tr, ts = timed(resident), timed(full)
knobs = {k: os.environ.get(k) for k in ('LP_OVERLAP_ENCODERS', 'LP_OVERLAP_CRITERIONS', 'LP_COPY_PRIORITY', 'GPU_MAX_HW_QUEUES') if os.environ.get(k) is not None}
knobs['SKIP_STREAMS'] = os.environ.get('SKIP_STREAMS', '0')
print(f'[input path] {wl} {knobs} host ms/step {last[0]}: resident {tr:.3f} ms/step, new host batch every step {ts:.3f} ms/step ({(ts / tr - 1) * 100:+.2f} %)', flush=True)
