"""Fused multi-tensor optimizers and EMA on the gfx950 kernels (lp_mt_optimizer_step / lp_mt_ema).

``FusedRAdam`` / ``FusedAdam`` are drop-ins for the reference's ``utils.radam.RAdam`` and ``torch.optim.Adam`` as configured
by runners/holycow.py:34-41 and discriminators/no_landmarks.py:26-28 (betas=(beta1, 0.999), eps=1e-5, no weight decay): same
constructor, same per-parameter state (``step``, ``exp_avg``, ``exp_avg_sq``) in ``state_dict()``, but ONE kernel launch per
``step()`` over all parameters and a device-resident step counter, so the whole training step can be captured in a hipGraph."""
import struct

import torch
from torch.optim.optimizer import Optimizer

from . import _lib
from ._lib import check


# Bumped by every in-place weight update that bypasses autograd's version counters (the fused optimizer / EMA kernels write through
# raw pointers): caches of derived data (the generator's inference weight packs) key on it.
WEIGHTS_GENERATION = [0]


def _build_table(entries, device):
    """entries: list of (p, g, m, v) tensors (m/v may be None) -> (device uint8 tensor with the packed MtDesc array, max numel)"""
    assert _lib.lib().lp_mt_desc_bytes() == 40
    blob = bytearray()
    max_n = 0
    for p, g, m, v in entries:
        for t in (p, g) + ((m, v) if m is not None else ()):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), 'fused optimizers need contiguous fp32 CUDA tensors'
        blob += struct.pack('<QQQQq', p.data_ptr(), g.data_ptr(), m.data_ptr() if m is not None else 0,
                            v.data_ptr() if v is not None else 0, p.numel())
        max_n = max(max_n, p.numel())
    table = torch.frombuffer(blob, dtype=torch.uint8).clone().to(device)
    return table, max_n


class _FusedBase(Optimizer):
    KIND = None

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, **_ignored):
        if weight_decay != 0:
            raise NotImplementedError('weight decay is not used by the reference configs and is not fused')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}
        self._parts = None          # set_partitions(): [(name, {id(param)})] + the implicit partition 'rest'
        self._stepped = set()

    def set_partitions(self, named):
        """Step named SUBSETS of the parameters with their own launches (round 6): ``step(part=name)`` updates that subset only -- e.g. the generator's
        parameters as soon as loss_G.backward has produced their gradients, on a side stream beside the encoders' backward -- and the next plain
        ``step()`` updates whatever has not been stepped since the previous plain ``step()``.  Every subset of a group keeps its own device step counter,
        advanced once per iteration like the group's single one: the update of every element is unchanged, bit for bit.  ``named``: {name: params}."""
        self._publish_steps()
        self._parts = [(name, {id(p) for p in ps}) for name, ps in named.items()]
        self._stepped = set()
        self._tables = {}

    def _part_names(self):
        return [None] if self._parts is None else [n for n, _ in self._parts] + ['rest']

    def _part_params(self, group, part):
        params = [p for p in group['params'] if p.requires_grad]
        if self._parts is None or part is None:
            return params
        if part == 'rest':
            taken = set().union(*[ids for _, ids in self._parts]) if self._parts else set()
            return [p for p in params if id(p) not in taken]
        ids = dict(self._parts)[part]
        return [p for p in params if id(p) in ids]

    def ensure_flat(self, gi=0):
        """Gradient arena: all gradients of a parameter group are views into ONE flat fp32 buffer (created once).  zero_grad is a
        single memset, the fused step addresses the views by pointer, and the data-parallel all-reduce (parallel.GradReducer)
        runs directly on the arena without gather/scatter copies."""
        flats = self.__dict__.setdefault('_flat', {})
        group = self.param_groups[gi]
        params = [p for p in group['params'] if p.requires_grad]
        key = tuple(p.data_ptr() for p in params)
        cur = flats.get(gi)
        if cur is None or cur[0] != key or any(p.grad is None or p.grad.data_ptr() != v.data_ptr() for p, v in zip(params, cur[2])):
            total = sum(p.numel() for p in params)
            flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
            views, off = [], 0
            for p in params:
                v = flat[off:off + p.numel()].view_as(p)
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v
                views.append(v)
                off += p.numel()
            cur = (key, flat, views)
            flats[gi] = cur
        return cur[1]

    def _prepare(self, gi, group, part=None):
        params = self._part_params(group, part)
        if not params:
            return None
        dev = params[0].device
        self.ensure_flat(gi)
        for p in params:
            st = self.state[p]
            if 'exp_avg' not in st:
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st['step'] = 0
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in params)
        tk = gi if part is None else (gi, part)
        cached = self._tables.get(tk)
        if cached is None or cached[0] != key:
            table, max_n = _build_table([(p.data, p.grad, self.state[p]['exp_avg'], self.state[p]['exp_avg_sq']) for p in params], dev)
            step0 = max((int(self.state[p]['step']) for p in params), default=0)
            step = cached[3] if cached is not None else torch.tensor([step0], dtype=torch.int64, device=dev)
            cached = (key, table, max_n, step, len(params))
            self._tables[tk] = cached
        return cached

    def zero_grad(self, set_to_none: bool = False):
        """gradients are zeroed IN PLACE (never released): the fused kernel addresses them by raw pointer -- one memset per group"""
        for gi, group in enumerate(self.param_groups):
            if any(p.requires_grad for p in group['params']):
                self.ensure_flat(gi).zero_()

    @torch.no_grad()
    def step(self, closure=None, part=None):
        """``part`` (after ``set_partitions``): update that subset only; a plain call updates every subset not stepped since the last plain call"""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from . import hipops as _ops
        _ops.sn_defer_check(type(self).__name__ + '.step')          #  .grad is incomplete while deferred spectral-norm jobs are pending
        if part is not None:
            assert self._parts is not None and part in self._part_names() and part not in self._stepped, (part, self._stepped)
            todo = [part]
        else:
            todo = [n for n in self._part_names() if n not in self._stepped]
        for gi, group in enumerate(self.param_groups):
            for name in todo:
                prep = self._prepare(gi, group, name)
                if prep is None:
                    continue
                _, table, max_n, step, n = prep
                b1, b2 = group['betas']
                check(_lib.lib().lp_mt_optimizer_step(table.data_ptr(), n, max_n, step.data_ptr(), self.KIND, group['lr'], b1, b2,
                                                      group['eps'], torch.cuda.current_stream().cuda_stream), 'lp_mt_optimizer_step')
        if part is not None:
            self._stepped.add(part)
        else:
            self._stepped = set()
        WEIGHTS_GENERATION[0] += 1
        return loss

    def _publish_steps(self):
        """device step counters -> the reference's per-parameter ``state[p]['step']``"""
        if not self._tables or torch.cuda.is_current_stream_capturing():      # (no table: nothing was stepped -- also the no-GPU case)
            return
        for tk, cached in self._tables.items():
            gi, part = (tk, None) if not isinstance(tk, tuple) else tk
            s = int(cached[3].item())
            for p in self._part_params(self.param_groups[gi], part):
                if p in self.state:
                    self.state[p]['step'] = s

    def state_dict(self):
        self._publish_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}


class FusedRAdam(_FusedBase):
    """utils/radam.py:29-95 (degenerated_to_sgd=True) as one launch per parameter group.  Differences from the reference class, all
    outside what its training loop does: every parameter of a group that requires grad is stepped on every call with ONE shared
    step counter (the reference skips a parameter whose ``.grad`` is None; here a gradient view always exists and is zero after
    ``zero_grad``, so such a parameter would still move on its momentum); ``state_dict()`` carries the reference's per-group
    ``buffer`` cache (unused here) so that the state loads into the reference's RAdam."""
    KIND = 0

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, **ignored):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **ignored)
        for group in self.param_groups:
            group.setdefault('buffer', [[None, None, None] for _ in range(10)])
        self.defaults.setdefault('buffer', [[None, None, None] for _ in range(10)])


class FusedAdam(_FusedBase):
    KIND = 1


class FusedEMA:
    """running averages of a module's parameters (alpha-blend) and buffers (copy) in two launches -- holycow.py:99-109"""

    def __init__(self, current: torch.nn.Module, average: torch.nn.Module):
        pc, pa = list(current.parameters()), list(average.parameters())
        bc, ba = list(current.buffers()), list(average.buffers())
        assert len(pc) == len(pa) and len(bc) == len(ba)
        self.keep = (pc, pa, bc, ba)
        dev = pc[0].device
        self.ptable, self.pmax = _build_table([(a.data, c.data, None, None) for c, a in zip(pc, pa)], dev)
        fb = [(a, c) for c, a in zip(bc, ba) if c.dtype == torch.float32 and c.numel() > 0]
        self.other = [(a, c) for c, a in zip(bc, ba) if c.dtype != torch.float32 and c.numel() > 0]     # e.g. BN num_batches_tracked
        self.btable, self.bmax = _build_table([(a.data, c.data, None, None) for a, c in fb], dev) if fb else (None, 0)
        self.np, self.nb = len(pc), len(fb)
        self.key = tuple(t.data_ptr() for t in pc + pa + bc + ba)

    def valid(self):
        pc, pa, bc, ba = self.keep
        return self.key == tuple(t.data_ptr() for t in pc + pa + bc + ba)

    @torch.no_grad()
    def update(self, alpha: float):
        st = torch.cuda.current_stream().cuda_stream
        WEIGHTS_GENERATION[0] += 1
        check(_lib.lib().lp_mt_ema(self.ptable.data_ptr(), self.np, self.pmax, alpha, 0, st), 'lp_mt_ema')
        if self.btable is not None:
            check(_lib.lib().lp_mt_ema(self.btable.data_ptr(), self.nb, self.bmax, 0.0, 1, st), 'lp_mt_ema')
        if self.other:                      # integer buffers (BatchNorm step counters): one multi-tensor copy, not one launch each
            torch._foreach_copy_([a for a, _ in self.other], [c for _, c in self.other])
