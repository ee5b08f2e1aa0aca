"""Concurrent branches of one training step on separate HIP streams.

A meta-training step issues ~2200 kernels, most of them far too small to fill 256 CUs (the MobileNetV2 pose encoder alone: ~600 launches
of 2 .. 20 us).  Branches that do not depend on each other -- the pose encoder beside the identity encoder, the VGG-19 and VGGFace
perceptual criterions beside the discriminator -- are therefore ISSUED ON SEPARATE STREAMS: ``fork`` makes a side stream wait for the
work already queued on the current one, ``join`` makes the current stream wait for the side stream.  Autograd runs every backward node
on the stream of its forward, so the backward passes of the branches overlap the same way; under hipGraph capture the cross-stream
waits become graph edges and the branches become parallel paths of the captured graph.  LP_OVERLAP=0 runs everything on one stream."""
import os
from contextlib import contextmanager

import torch

_STREAMS = {}


# measured on the captured steps (profiles/r03_stream_overlap.txt): meta-training 43.4 ms on one stream, 40.8 with the encoders side by side,
# 39.0 with the VGG criterions beside the discriminator as well; the fine-tuning step (no identity encoder, pose encoder without autograd)
# only has the criterion overlap to offer and loses 1.6 % with it.  optimizer_G.step + EMA beside the discriminator backward: 38.98 -> 38.87 ms
# (meta-training, within noise) and 22.47 -> 23.17 ms (fine-tuning): off by default.  The identity encoder's weight gradients beside its
# data-gradient chain: 38.8 -> 41.7 ms (two bandwidth-bound kernel sequences sharing HBM and the L2s run slower than one after the other): off.
# The target-image halves of the VGG criterions at the top of the step (beside encoders + generator): 38.9 -> 42.3 ms (meta-training),
# 22.4 -> 24.1 ms (fine-tuning) -- large kernels beside large kernels again: off.  What pays is a branch of SMALL kernels beside a
# branch of large ones -- e.g. the spectral-norm power iterations + weight packs of G and D (~40 short launches) beside the encoders:
# 38.95 -> 38.62 ms, on (meta-training); the discriminator's three passes beside each other (forward and, through autograd, the whole of
# loss_D.backward): 38.9 -> 37.4 ms, on (meta-training; fine-tuning 22.34 -> 22.09 ms, left off: that step stays a single-stream graph).  LP_OVERLAP_{ENCODERS,CRITERIONS,PREPARE,DPASSES,REAL,OPTIMIZER,WGRAD,TARGETS,EBWD} = 0 | 1 force; LP_OVERLAP=0 turns
# everything off.
def enabled(t, what: str, finetuning: bool = False) -> bool:
    """``what``: 'encoders' (pose encoder beside the identity encoder) | 'criterions' (VGG stacks beside the discriminator pass, their
    target-image halves beside encoders + generator) | 'optimizer' (optimizer_G.step + EMA beside the discriminator backward) |
    'wgrad' (the identity encoder's weight gradients beside its data-gradient chain) | 'targets' (only the target-image halves of the
    VGG criterions ahead of encoders + generator) | 'prepare' (spectral-norm power iterations + weight packs of G and D beside the encoders) | 'dpasses' (the discriminator's three
    passes beside each other) | 'real' (with 'dpasses' + 'prepare': its real-image pass already beside the generator's forward) | 'ebwd' (one GPU, meta-training: the encoders' backward beside loss_D.backward -- runners/holycow.py cuts the
    autograd graph behind the embedder)"""
    if not (torch.is_tensor(t) and t.is_cuda) or os.environ.get('LP_OVERLAP', '1') == '0':
        return False
    # 'ebwd': measured null in round 4 (42.9 / 43.5 ms on vs 43.4 / 43.5 ms off, profiles/r04_stream_overlap.txt: both backward passes bound by HBM
    # traffic with an all-fp16 critic)
    # 'real' (round 4): the critic's pass over the REAL image issued beside the generator's forward (whose 4x4 .. 32x32 layers leave most of the
    # chip idle) instead of beside the other two passes: 42.2 -> 41.75 ms; the VGG target halves at the same place ('targets' = 2): 42.2 -> 42.0,
    # not additive (profiles/r04_stream_overlap.txt)
    # round 6: with the generator and the critic's D-side tail on bf16x3 operands the discriminator-side backward is matrix-bound enough to run
    # beside the encoders' bandwidth-bound backward: 43.11 / 43.09 / 43.24 ms off vs 42.71 / 42.06 / 42.62 ms on (profiles/r06_stream_overlap.txt) -- ON
    # round 6: the FINE-TUNING step takes the same branches (it stayed a single-stream graph through round 5: -1.6 % / +1.1 % for single branches with
    # fp16 operands).  With the bf16x3 generator / critic tail: 22.47 ms one stream, 22.16 dpasses, 21.35 + prepare, 21.39 + real, 19.63 ms + criterions
    # (profiles/r06_stream_overlap.txt) -- on.  'ebwd' has no meaning there (no encoder is trained).
    # 'optimizer' (optimizer_G.step + EMA beside loss_D.backward, graphed step): meta-training +-0 (42.51 vs 42.52 ms), fine-tuning 19.75 -> 19.51 ms -- on there
    default = '0' if (what in ('wgrad', 'targets', 'gwgrad') or (what == 'optimizer' and not finetuning) or (what == 'ebwd' and finetuning)) else '1'
    return os.environ.get('LP_OVERLAP_' + what.upper(), default) != '0'


def side_stream(device, index: int) -> 'torch.cuda.Stream':
    key = (torch.device(device).index or 0, index)
    if key not in _STREAMS:
        if not _STREAMS:
            # a parameter whose backward node runs on a side stream is accumulated there while its AccumulateGrad node was created on
            # the main stream: intended here (the engine orders the two streams), so the per-step warning about it is switched off
            quiet = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
            if quiet is not None:
                quiet(False)
        _STREAMS[key] = torch.cuda.Stream(device=device)
    return _STREAMS[key]


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif isinstance(obj, (tuple, list)):
        for v in obj:
            yield from _tensors(v)


def join_all(device=None):
    """make the current stream wait for every side stream (of ``device``) that has taken part so far.  Called after a backward pass: a
    branch whose backward feeds nothing downstream (the discriminator's detached-input pass: its gradients are accumulated inside its own
    kernels) is otherwise never waited for by autograd -- harmless in eager mode only until the next consumer of those gradients runs on
    another stream, and an 'unjoined work' error at the end of a hipGraph capture.  Under capture only side streams that are themselves
    part of the capture are waited for (waiting for an uncaptured stream would be an illegal dependency)."""
    if not _STREAMS:
        return
    main = torch.cuda.current_stream(device)
    dev = main.device.index or 0
    capturing = torch.cuda.is_current_stream_capturing()
    for (d, _), side in _STREAMS.items():
        if d != dev or side == main:          # (called from inside a branch: a stream does not wait for itself)
            continue
        if capturing:
            with torch.cuda.stream(side):
                if not torch.cuda.is_current_stream_capturing():
                    continue
        main.wait_stream(side)


def fork_point(device):
    """an event at the current position of the current stream: ``branch(..., after=event)`` lets a side stream start from HERE even though
    more work has been queued on the current stream in the meantime (the host issues the branches one after the other)"""
    return torch.cuda.current_stream(device).record_event()


@contextmanager
def branch(device, index: int, after=None):
    """``with branch(dev, i) as b: out = f(...)`` runs f on side stream i, ordered after everything queued on the current stream so far
    (or after the ``fork_point`` event ``after``); call ``b.join(out)`` (any time later, on the original stream) before the results are
    used there."""
    main = torch.cuda.current_stream(device)
    side = side_stream(device, index)
    if after is not None:
        side.wait_event(after)
    else:
        side.wait_stream(main)
    b = _Branch(main, side)
    with torch.cuda.stream(side):
        yield b


class _Branch:
    def __init__(self, main, side):
        self.main, self.side = main, side

    def join(self, outputs=None):
        self.main.wait_stream(self.side)
        for t in _tensors(outputs):          # allocated on the side stream, consumed on the main one: keep the block until that work is done
            if t.is_cuda:
                t.record_stream(self.main)
