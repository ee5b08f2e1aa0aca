"""(f)4 input path (dataloaders/prefetch.py): the double-buffered pinned H2D prefetcher delivers exactly the loader's batches with no
overwrite hazard while the device is busy, and feeding the captured training step a NEW host batch every iteration costs <= 3 % over
replaying a resident batch (reference: dataloaders/dataloader.py:24-50 + the blocking dict_to_device of runners/holycow.py:233-236)."""
import os
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))
pytestmark = pytest.mark.gpu


def test_prefetcher_delivers_every_batch_while_the_device_is_busy():
    from latent_pose_reenactment_amd.dataloaders.prefetch import DevicePrefetcher
    g = torch.Generator().manual_seed(0)
    batches = [({'x': torch.randn(4, 3, 256, 256, generator=g), 'tag': k}, {'label': torch.arange(4) + k}) for k in range(9)]
    batches[3][0]['x'] = batches[3][0]['x'].pin_memory()          # a loader that already delivers pinned tensors: copied from in place
    big = torch.randn(4096, 4096, device='cuda')
    got = []
    for data, target in DevicePrefetcher(batches, 'cuda'):
        assert data['x'].is_cuda and target['label'].is_cuda and isinstance(data['tag'], int)
        acc = big
        for _ in range(6):                       # ~10 ms of device work enqueued BEFORE the batch is read: a premature refill of the
            acc = (acc @ big) * 1e-3             # slot (two batches later) would be caught by the checksum below
        got.append((data['x'].double().sum() + acc[0, 0] * 0, target['label'].clone(), data['tag']))
    torch.cuda.synchronize()
    assert len(got) == len(batches)
    for (s, lab, tag), (d, t) in zip(got, batches):
        assert tag == d['tag'] and torch.equal(lab.cpu(), t['label'])
        assert abs(float(s) - float(d['x'].double().sum())) < 1e-6 * d['x'].numel()


def test_new_host_batch_every_step_costs_at_most_three_percent(monkeypatch):
    """the captured fine-tuning step (BASELINE configs[1], bs 8, 256 x 256): resident batch vs a fresh host batch per iteration through
    the prefetcher + GraphedTrainStep.load_batch (device-to-device into the static inputs)"""
    import bench
    from latent_pose_reenactment_amd.dataloaders.prefetch import DevicePrefetcher
    args = bench.make_args(256, 8, 'cuda:0', 1, 0, 'f16', finetune=True)
    args.generator = 'vector_pose_unsupervised_segmentation_noBottleneck'
    tm, opt_G, opt_D, holycow = bench.build(args)
    data, target = bench.synthetic_batch(args, 8, seed=123)
    step = holycow.GraphedTrainStep(tm, opt_G, opt_D, args, data, target, warmup_steps=3)
    host = [({k: (v.cpu() + 0.001 * i if v.is_floating_point() else v.cpu()) for k, v in data.items()},
             {k: v.cpu() for k, v in target.items()}) for i in range(4)]

    def resident(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

    def streamed(n):
        loader = (host[i % len(host)] for i in range(n))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for d, t in DevicePrefetcher(loader, 'cuda:0'):
            step.load_batch(d, t)
            step()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
    resident(5); streamed(5)
    tr = min(resident(30) for _ in range(3))
    ts = min(streamed(30) for _ in range(3))
    print(f'[input path] resident batch {tr * 1e3:.3f} ms/step, new pinned host batch every step {ts * 1e3:.3f} ms/step ({(ts / tr - 1) * 100:+.2f} %)')
    # (round 6: the fine-tuning step runs its branches on several streams and dropped from 21.4 to 19.4 ms; the staged copy beside it costs
    #  0.43 ms = +2.2 % where it cost 0.1 ms = +0.5 % beside the one-stream step -- gate 3 %)
    assert ts <= tr * 1.03, (tr, ts)
    # and the step really consumed the streamed data: the static input now holds the last host batch
    last = host[(30 - 1) % len(host)][0]['target_rgbs']
    assert torch.allclose(step.data['target_rgbs'].cpu(), last)
