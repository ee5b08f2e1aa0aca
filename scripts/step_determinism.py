"""Run-to-run repeatability of the training step inside ONE process: the same model (same seed) and the same batch are stepped twice from
scratch; every parameter and buffer of the critic is compared bit for bit after each step.  (The critic's arithmetic has no float atomics; the
generator / encoder gradients pass through the crop-and-resize backward's atomics and may differ in the last bits.)  Used to separate a
data-parallel exchange problem from nondeterminism of the step itself: scripts/dp_replica_diag.py found replicas whose spectral-norm (u, v)
buffers differ although their weights are bit-identical.
usage: python scripts/step_determinism.py [eager|graph] [steps=3] [image_size=128]   (LP_OVERLAP=0 for the one-stream step)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))
import torch  # noqa: E402

import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'eager'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
size = int(sys.argv[3]) if len(sys.argv) > 3 else 128


def run():
    args = bench.make_args(size, 8, 'cuda:0', 1, 0, os.environ.get('LP_PREC', 'f16'), finetune=False)
    tm, opt_G, opt_D, holycow = bench.build(args)
    data, target = bench.synthetic_batch(args, 8, seed=123)
    snaps = []
    if mode == 'graph':
        step = holycow.GraphedTrainStep(tm, opt_G, opt_D, args, data, target, warmup_steps=1)
    else:
        def step():
            holycow.train_step(tm, data, target, opt_G, opt_D, args)
    for _ in range(steps):
        step()
        torch.cuda.synchronize()
        snaps.append({k: v.detach().clone() for k, v in tm.discriminator.state_dict().items() if v.dtype == torch.float32})
    return snaps


a = run()
b = run()
for i, (sa, sb) in enumerate(zip(a, b)):
    bad = [(k, float((sa[k] - sb[k]).abs().max())) for k in sa if not torch.equal(sa[k].view(torch.int32), sb[k].view(torch.int32))]
    print(f'[step-determinism] {mode} LP_OVERLAP={os.environ.get("LP_OVERLAP", "default")}: after step {i + 1}: {len(bad)} of {len(sa)} critic tensors differ between two runs'
          + (': ' + ', '.join(f'{k} ({d:.1e})' for k, d in bad[:10]) if bad else ''), flush=True)
