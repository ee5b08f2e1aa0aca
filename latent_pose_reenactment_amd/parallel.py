"""Data-parallel gradient exchange over RCCL/xGMI -- replaces the reference's apex ``Reducer`` (train.py:196-200,
runners/holycow.py:241-242,249-250).

The reference flattens EVERY parameter gradient of embedder+generator+discriminator and all-reduces the lot twice per
iteration.  Result-identical and cheaper: after ``loss_G.backward`` only the embedder/generator gradients
are consumed (the discriminator's are zeroed before its own backward), after ``loss_D.backward`` only the
discriminator's.  Each side is one flat fp32 bucket (one large RCCL all-reduce: xGMI is per-link bound, fewer/larger
collectives win), averaged by world size, copied back.  ``async_op`` lets the G-side reduction overlap the D backward.
Works with the ``gloo`` backend on CPU (tests) and ``nccl`` (= RCCL) on MI355X."""
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params: Iterable[torch.nn.Parameter], optimizer=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.flat: Optional[torch.Tensor] = None
        self.handle = None
        self.live: List[torch.nn.Parameter] = []
        self.optimizer = optimizer if hasattr(optimizer, 'ensure_flat') else None
        self.arena = None

    def start(self, world_size: int, async_op: bool):
        if self.optimizer is not None and len(self.optimizer.param_groups) == 1:
            # fused optimizers keep every gradient in one flat arena: reduce it in place, no gather/scatter
            self.arena = self.optimizer.ensure_flat(0)
            self.handle = dist.all_reduce(self.arena, op=dist.ReduceOp.SUM, async_op=async_op)
            if not async_op:
                self.finish(world_size)
            return
        self.live = [p for p in self.params if p.grad is not None]
        if not self.live:
            return
        n = sum(p.grad.numel() for p in self.live)
        if self.flat is None or self.flat.numel() != n or self.flat.device != self.live[0].grad.device:
            self.flat = torch.empty(n, dtype=torch.float32, device=self.live[0].grad.device)
        torch.cat([p.grad.reshape(-1) for p in self.live], out=self.flat)      # one gather kernel instead of one copy per tensor
        self.handle = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if not async_op:
            self.finish(world_size)

    def finish(self, world_size: int):
        if self.arena is not None:
            if self.handle is not None and hasattr(self.handle, 'wait'):
                self.handle.wait()
            self.handle = None
            self.arena.div_(world_size)
            self.arena = None
            return
        if not self.live:
            return
        if self.handle is not None:
            self.handle.wait()
            self.handle = None
        self.flat.div_(world_size)
        views, off = [], 0
        for p in self.live:
            k = p.grad.numel()
            views.append(self.flat[off:off + k].view_as(p.grad))
            off += k
        torch._foreach_copy_([p.grad for p in self.live], views)              # one multi-tensor scatter
        self.live = []


class GradReducer:
    def __init__(self, training_module, finetune: bool = False, broadcast: bool = True, optimizer_G=None, optimizer_D=None):
        """``optimizer_G/_D`` (optional): the fused optimizers; their flat gradient arenas are then all-reduced in place."""
        self.world_size = dist.get_world_size()
        g_side = list(training_module.generator.parameters())
        if not finetune:
            g_side += list(training_module.embedder.parameters())
        self.g_bucket = _Bucket(g_side, optimizer_G)
        self.d_bucket = _Bucket(training_module.discriminator.parameters(), optimizer_D)
        if broadcast:        # apex Reducer broadcasts rank 0's parameters at construction
            with torch.no_grad():
                for t in training_module.parameters():      # parameters only, like apex (buffers/EMA stay rank-local)
                    dist.broadcast(t, 0)

    def reduce_generator_side(self, async_op: bool = False):
        self.g_bucket.start(self.world_size, async_op)

    def wait_generator_side(self):
        self.g_bucket.finish(self.world_size)

    def reduce_discriminator_side(self, async_op: bool = False):
        self.d_bucket.start(self.world_size, async_op)

    def wait_discriminator_side(self):
        self.d_bucket.finish(self.world_size)

    def reduce(self):
        """apex-compatible entry point: reduce everything that currently has a gradient"""
        self.reduce_generator_side()
        self.reduce_discriminator_side()
