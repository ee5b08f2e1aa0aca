"""Data-parallel gradient exchange over RCCL/xGMI -- replaces the reference's apex ``Reducer`` (train.py:196-200,
runners/holycow.py:241-242,249-250).

The reference flattens EVERY parameter gradient of embedder+generator+discriminator and all-reduces the lot twice per
iteration.  Result-identical and cheaper: after ``loss_G.backward`` only the embedder/generator gradients
are consumed (the discriminator's are zeroed before its own backward), after ``loss_D.backward`` only the
discriminator's.  Each side is one flat fp32 bucket (one large RCCL all-reduce: xGMI is per-link bound, fewer/larger
collectives win), averaged by world size, copied back.  ``async_op`` lets the G-side reduction overlap the D backward
(runners/holycow.py issues it on RCCL's own stream right after loss_G.backward and waits for it before optimizer_G.step).

Meta-training: 50.2 M of the discriminator's 69.7 M gradient elements belong to the 98000 x 512 label embedding, of which only
the <= B rows of this rank's labels are non-zero apart from a rank-1 term (nn.SNEmbeddingFn).  Instead of all-reducing that dense
200 MB slice, every rank publishes (labels, its B gradient rows, the rank-1 coefficient) -- B x 513 + 1 floats -- and rebuilds the
averaged gradient locally: identical arithmetic in identical order on every rank, so the replicas stay bit-identical.
Works with the ``gloo`` backend (tests; gathers are expressed as all-reduces of zero-padded buffers because gloo has no
all_gather for device tensors) and ``nccl`` (= RCCL) on MI355X."""
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def flat_broadcast(tensors, src: int = 0):
    """broadcast many tensors from ``src`` with one collective per (dtype, device): gather into a flat buffer, broadcast, scatter
    back with a multi-tensor copy"""
    groups = {}
    for t in tensors:
        groups.setdefault((t.dtype, t.device), []).append(t)
    with torch.no_grad():
        for ts in groups.values():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src)
            views, off = [], 0
            for t in ts:
                views.append(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
            torch._foreach_copy_(ts, views)


def _mean_op():
    """(reduce op, needs a divide afterwards): RCCL averages inside the collective (ReduceOp.AVG: no extra pass over a 256 MB arena);
    gloo (functional tests only) has no AVG -- SUM, then one ``div_``"""
    if dist.get_backend() == 'nccl':
        return dist.ReduceOp.AVG, False
    return dist.ReduceOp.SUM, True


class _Bucket:
    def __init__(self, params: Iterable[torch.nn.Parameter], optimizer=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.flat: Optional[torch.Tensor] = None
        self.handle = None
        self.parts: List[torch.Tensor] = []
        self.live: List[torch.nn.Parameter] = []
        self.optimizer = optimizer if hasattr(optimizer, 'ensure_flat') else None
        self.arena = None
        self.divide = False
        self.diag = None          # GradReducer.enable_diag(): list of per-collective records (bytes, issue / wait events)
        self.name = ''

    def _diag_issue(self, parts, tag=None):
        if self.diag is None or not parts or not parts[0].is_cuda:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        rec = {'bucket': tag or self.name, 'bytes': int(sum(t.numel() * t.element_size() for t in parts)), 'issue': ev}
        self.diag.append(rec)
        return rec

    def _diag_wait(self, begin):
        """brackets the wait of ``finish``: the span between the two events is the time the COMPUTE stream stood still for the exchange"""
        if self.diag is None or not torch.cuda.is_available():
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        for rec in self.diag:
            if rec['bucket'].split(':')[0] == self.name and ('done' if not begin else 'wait') not in rec:
                rec['wait' if begin else 'done'] = ev
        return ev

    def arena_slice_of(self, param):
        """(offset, numel) of ``param``'s gradient inside the optimizer's flat arena, or None"""
        if self.optimizer is None or len(self.optimizer.param_groups) != 1:
            return None
        off = 0
        for p in self.optimizer.param_groups[0]['params']:
            if not p.requires_grad:
                continue
            if p is param:
                return off, p.numel()
            off += p.numel()
        return None

    def start(self, world_size: int, async_op: bool, skip=None, only=None):
        """``skip`` = (offset, numel): leave that slice of the arena out of the all-reduce (the caller rebuilds it); ``only`` = (offset,
        numel): all-reduce just that slice now -- further ``start`` calls add more slices, ``finish`` waits for all of them (the
        generator-side arena goes out as two buckets: the generator's slice before the encoders' backward, the encoders' slice after)"""
        op, self.divide = _mean_op()
        from . import hipops as _ops
        _ops.sn_defer_check('GradReducer')          #  no exchange of gradients that a deferred spectral-norm job has yet to complete
        if self.optimizer is not None and len(self.optimizer.param_groups) == 1:
            # fused optimizers keep every gradient in one flat arena: reduce it in place, no gather/scatter
            arena = self.optimizer.ensure_flat(0)
            if self.arena is None:
                self.arena, self.handle, self.parts = arena, [], []
            if only is not None:
                off, n = only
                parts = [arena[off:off + n]]
            elif skip is not None:
                off, n = skip
                parts = [arena[:off], arena[off + n:]]
            else:
                parts = [arena]
            parts = [t for t in parts if t.numel()]
            self.parts += parts
            self._diag_issue(parts, None if only is None else f'{self.name}:{only[0]}+{only[1]}')
            self.handle += [dist.all_reduce(t, op=op, async_op=async_op) for t in parts]
            if not async_op:
                self.finish(world_size)
            return
        assert only is None and skip is None, 'slices of a bucket need the fused optimizers\' gradient arena'
        self.live = [p for p in self.params if p.grad is not None]
        if not self.live:
            return
        n = sum(p.grad.numel() for p in self.live)
        if self.flat is None or self.flat.numel() != n or self.flat.device != self.live[0].grad.device:
            self.flat = torch.empty(n, dtype=torch.float32, device=self.live[0].grad.device)
        torch.cat([p.grad.reshape(-1) for p in self.live], out=self.flat)      # one gather kernel instead of one copy per tensor
        self._diag_issue([self.flat])
        self.handle = dist.all_reduce(self.flat, op=op, async_op=async_op)
        if not async_op:
            self.finish(world_size)

    def finish(self, world_size: int):
        if self.arena is not None:
            self._diag_wait(True)
            for h in (self.handle or []):
                if h is not None and hasattr(h, 'wait'):
                    h.wait()          # (RCCL: the current stream waits for the collective's stream; the host does not block)
            self._diag_wait(False)
            self.handle = None
            if self.divide:
                for t in self.parts:
                    t.div_(world_size)
            self.arena, self.parts = None, []
            return
        if not self.live:
            return
        if self.handle is not None:
            self._diag_wait(True)
            self.handle.wait()
            self._diag_wait(False)
            self.handle = None
        if self.divide:
            self.flat.div_(world_size)
        views, off = [], 0
        for p in self.live:
            k = p.grad.numel()
            views.append(self.flat[off:off + k].view_as(p.grad))
            off += k
        torch._foreach_copy_([p.grad for p in self.live], views)              # one multi-tensor scatter
        self.live = []


class GradReducer:
    def __init__(self, training_module, finetune: bool = False, broadcast: bool = True, optimizer_G=None, optimizer_D=None,
                 max_batch: Optional[int] = None):
        """``optimizer_G/_D`` (optional): the fused optimizers; their flat gradient arenas are then all-reduced in place.
        ``max_batch``: the largest per-rank batch (rows of the label-embedding exchange); None = agreed on at the first exchange."""
        self.world_size = dist.get_world_size()
        gen = [p for p in training_module.generator.parameters()]
        emb = [] if finetune else [p for p in training_module.embedder.parameters()]
        self.g_bucket = _Bucket(gen + emb, optimizer_G)
        self.g_bucket.name = 'G-side arena (generator | encoders)'
        self._arena_layout_checked = False
        # the generator-side arena is [generator | embedder] (runners/holycow.get_optimizer): two buckets of ONE buffer
        self.g_parts = {'generator': _Bucket(gen), 'embedder': _Bucket(emb)}           # (plain-parameter path: one flat buffer each)
        self.n_gen = sum(p.numel() for p in gen if p.requires_grad)
        self.n_emb = sum(p.numel() for p in emb if p.requires_grad)
        self._g_split_live = []
        self.g_parts['generator'].name, self.g_parts['embedder'].name = 'generator', 'encoders'
        self.d_bucket = _Bucket(training_module.discriminator.parameters(), optimizer_D)
        self.d_bucket.name = 'D-side arena (critic without the label embedding)'
        self.discriminator = training_module.discriminator
        self.max_batch = max_batch
        self.use_sparse = None          # agreed on by all ranks at the first discriminator-side exchange
        if broadcast:        # apex Reducer broadcasts rank 0's parameters at construction
            # ONE flat collective per dtype instead of one per tensor (hundreds of tiny broadcasts).  Parameters only, like apex
            # (buffers / EMA stay rank-local) -- plus the (u, v) power-iteration buffers of the label embedding: the row-sparse
            # exchange below rebuilds the rank-1 term of its gradient from THIS rank's u, v, which is only the averaged gradient if
            # every rank holds the same vectors (same start + identical weights after every all-reduce => they stay identical).
            tensors = [t.data for t in training_module.parameters()]
            emb_l = getattr(training_module.discriminator, 'embed', None)
            if emb_l is not None and hasattr(emb_l, 'weight_u'):
                tensors += [emb_l.weight_u, emb_l.weight_v]
            flat_broadcast(tensors, 0)

    def reduce_generator_side(self, async_op: bool = False, part: Optional[str] = None):
        """``part`` None: the whole generator-side gradient set in one all-reduce.  'generator' | 'embedder': that bucket only -- the
        generator's 150 MB of gradients are final when ``loss_G.backward`` reaches the embedder's outputs, i.e. BEFORE the encoders'
        backward (>= 10 ms of kernels) starts: runners/holycow.py cuts the backward pass there and issues the generator bucket first, the
        encoders' bucket when their backward is done.  ``wait_generator_side`` waits for everything issued."""
        if part is None:
            self.g_bucket.start(self.world_size, async_op)
            return
        assert part in ('generator', 'embedder'), part
        if self.g_bucket.optimizer is not None and len(self.g_bucket.optimizer.param_groups) == 1:
            only = self._g_arena_slices()[part]
            if only[1]:
                self.g_bucket.start(self.world_size, async_op, only=only)
            return
        b = self.g_parts[part]
        b.start(self.world_size, async_op)
        if async_op:
            self._g_split_live.append(b)

    def _g_arena_slices(self):
        """(offset, numel) of the generator's and of the encoders' gradients inside optimizer_G's flat arena, DERIVED from the optimizer's own
        parameter list: each must be one contiguous run, the two must tile the arena -- a shared / tied parameter, another
        get_optimizer ordering or an extra parameter would otherwise leave part of the arena un-averaged without any error"""
        cached = self.__dict__.get('_g_slices')
        if cached is not None:
            return cached
        gen_ids = {id(p) for p in self.g_parts['generator'].params}
        emb_ids = {id(p) for p in self.g_parts['embedder'].params}
        runs, off = [], 0
        for p in self.g_bucket.optimizer.param_groups[0]['params']:
            if not p.requires_grad:
                continue
            owner = 'generator' if id(p) in gen_ids else 'embedder' if id(p) in emb_ids else None
            if owner is None:
                raise RuntimeError('GradReducer: optimizer_G holds a parameter that belongs to neither the generator nor the embedder')
            if runs and runs[-1][0] == owner:
                runs[-1][2] += p.numel()
            else:
                runs.append([owner, off, p.numel()])
            off += p.numel()
        owners = [r[0] for r in runs]
        if len(owners) != len(set(owners)):
            raise RuntimeError(f'GradReducer: generator and encoder parameters interleave in optimizer_G ({owners}); the two-bucket exchange needs one contiguous run each')
        out = {'generator': (0, 0), 'embedder': (0, 0)}
        for owner, o, n in runs:
            out[owner] = (o, n)
        if out['generator'][1] != self.n_gen or out['embedder'][1] != self.n_emb or self.n_gen + self.n_emb != off:
            raise RuntimeError(f'GradReducer: optimizer_G covers {off} gradient elements in runs {runs}, expected generator {self.n_gen} + encoders {self.n_emb}')
        self.__dict__['_g_slices'] = out
        return out

    # ---- diagnosis of the exchange (bench.py --gpus N prints it: the first run on real xGMI then yields a diagnosis, not just a number) --------
    def enable_diag(self):
        """record, per collective issued from now on: bytes, an event at its issue and the events around the wait of its consumer"""
        self.diag = []
        for b in [self.g_bucket, self.d_bucket] + list(self.g_parts.values()):
            b.diag = self.diag

    def diag_summary(self, steps: int):
        """-> per bucket: bytes per step, mean issue -> consumer-wait span and the EXPOSED time (what the compute stream actually stood still
        in ``finish``), plus ``exposed_comm_ms`` per step.  Call after a device synchronize."""
        if not getattr(self, 'diag', None):
            return None
        agg = {}
        for rec in self.diag:
            if 'wait' not in rec or 'done' not in rec:
                continue
            a = agg.setdefault(rec['bucket'], {'bytes': 0, 'n': 0, 'issue_to_wait_ms': 0.0, 'exposed_ms': 0.0})
            a['bytes'] += rec['bytes']; a['n'] += 1
            a['issue_to_wait_ms'] += rec['issue'].elapsed_time(rec['wait'])
            a['exposed_ms'] += rec['wait'].elapsed_time(rec['done'])
        out = {'buckets': {k: {'all_reduce_bytes_per_step': v['bytes'] // max(steps, 1), 'collectives_per_step': v['n'] / max(steps, 1),
                               'issue_to_consumer_wait_ms': round(v['issue_to_wait_ms'] / max(v['n'], 1), 3),
                               'exposed_wait_ms': round(v['exposed_ms'] / max(v['n'], 1), 3)} for k, v in agg.items()}}
        # the wait events of the slices of one arena coincide (one ``finish``): count each arena's wait once per step
        seen, total = set(), 0.0
        for rec in self.diag:
            if 'wait' in rec and 'done' in rec and id(rec['wait']) not in seen:
                seen.add(id(rec['wait']))
                total += rec['wait'].elapsed_time(rec['done'])
        out['exposed_comm_ms_per_step'] = round(total / max(steps, 1), 3)
        out['note'] = ('issue_to_consumer_wait: time the collective had to complete behind other work before its optimizer step asked for it; exposed_wait: '
                       'how long the compute stream then still stood in the wait (0 = fully hidden); the label-embedding rows go through two small '
                       'padded all-reduces that are not listed')
        return out

    def wait_generator_side(self):
        self.g_bucket.finish(self.world_size)
        for b in self._g_split_live:
            b.finish(self.world_size)
        self._g_split_live = []

    def reduce_discriminator_side(self, async_op: bool = False):
        """All-reduce (mean) of the discriminator arena; the label-embedding slice is rebuilt from a row-sparse exchange.

        The SEQUENCE AND SIZES of the collectives issued here depend only on state every rank shares: at the FIRST
        call every rank -- whether or not it has sparse parts to publish -- takes part in ONE int64 all-reduce that agrees on (a) whether
        ALL ranks use the row-sparse exchange (MIN of the local flags) and (b) its capacity in rows per rank (MAX of the local batch sizes,
        unless ``max_batch`` was given to the constructor).  Afterwards each rank publishes its own row count next to its labels, so a
        ragged last batch on some ranks (b < capacity) needs no other collective: its unused rows are zero and add nothing.
        Issue order: the two small exchanges first, then the arena all-reduce ASYNCHRONOUSLY on RCCL's stream, then the rebuild of the
        embedding gradient on the compute stream while the arena is in flight; ``finish`` waits."""
        sparse = self._sparse_embedding()
        if self.use_sparse is None:
            dev = next(iter(self.discriminator.parameters())).device
            b_local = int(sparse[1][1].shape[0]) if sparse is not None else 0
            agree = torch.tensor([1 if sparse is not None else 0, -(self.max_batch if self.max_batch is not None else b_local)], dtype=torch.int64, device=dev)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)          # MIN of the flags; MIN of the negated sizes = MAX of the sizes
            self.use_sparse = bool(int(agree[0].item()))
            if self.max_batch is None and self.use_sparse:
                self.max_batch = -int(agree[1].item())
        if not self.use_sparse:
            self.d_bucket.start(self.world_size, async_op)
            return
        if sparse is None:
            raise RuntimeError('GradReducer: the row-sparse label-embedding exchange was agreed on at the first step, but this rank has no '
                               'gradient parts to publish now (every rank must run the same discriminator backward)')
        (off, n), (label, rows, coef, u, v) = sparse
        b, e = rows.shape
        cap = self.max_batch
        if b > cap:
            raise RuntimeError(f'GradReducer: this rank brought {b} samples but the row-sparse exchange was sized for {cap} per rank at its '
                               'first call; construct GradReducer(max_batch=<largest per-GPU batch>)')
        rank = dist.get_rank()
        # fp32 [world, cap * E + 1]: gradient rows + rank-1 coefficient; int64 [world, cap]: the labels (exact for any label value; the
        # padding rows carry label 0 and zero gradient rows).  Zero-padded all-reduces: gloo has no all_gather for device tensors.
        buf = torch.zeros(self.world_size, cap * e + 1, dtype=rows.dtype, device=rows.device)
        buf[rank, :b * e] = rows.reshape(-1)
        buf[rank, cap * e] = coef.reshape(())
        lab = torch.zeros(self.world_size, cap, dtype=torch.int64, device=rows.device)
        lab[rank, :b] = label.to(torch.int64)
        h1 = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
        h2 = dist.all_reduce(lab, op=dist.ReduceOp.SUM, async_op=True)
        self.d_bucket.start(self.world_size, True, skip=(off, n))
        h1.wait(); h2.wait()          # (RCCL: stream dependencies, the host does not block; the arena all-reduce stays in flight)
        grad = self.d_bucket.arena[off:off + n].view(-1, e)
        inv = 1.0 / self.world_size
        grad.zero_()
        coef_mean = (buf[:, -1].sum() * inv).reshape(1)
        all_rows = (buf[:, :cap * e] * inv).reshape(self.world_size * cap, e)          # rank-major: the fixed order every rank rebuilds in
        if grad.is_cuda and e % 4 == 0:
            # the same launch pair SNEmbeddingFn.backward uses at N = 1 (lp_sn_embed_grad: rank-1 term, then the rows in order) -- no library GEMM
            from . import _lib
            _lib.check(_lib.lib().lp_sn_embed_grad(grad.data_ptr(), u.contiguous().data_ptr(), v.contiguous().data_ptr(), coef_mean.contiguous().data_ptr(),
                                                   lab.reshape(-1).contiguous().data_ptr(), all_rows.contiguous().data_ptr(), grad.shape[0], e,
                                                   self.world_size * cap, torch.cuda.current_stream().cuda_stream), 'lp_sn_embed_grad')
        else:       # CPU tensors: the gloo host-logic tests of this exchange (tests/test_host_logic.py); the product's gradients live on the GPU
            grad.addmm_((u * (-coef_mean))[:, None], v[None, :])
            grad.index_add_(0, lab.reshape(-1), all_rows)
        if not async_op:
            self.d_bucket.finish(self.world_size)

    def _sparse_embedding(self):
        """((offset, numel) of the label embedding's gradient in the discriminator arena, parts of this rank's gradient) when the
        row-sparse exchange applies: meta-training (many labels), fused optimizer arena, parts published by the last backward"""
        emb = getattr(self.discriminator, 'embed', None)
        parts = getattr(self.discriminator, '_embed_parts', {}).get('parts')
        if emb is None or parts is None or emb.weight_orig.shape[0] <= 4096:
            return None
        sl = self.d_bucket.arena_slice_of(emb.weight_orig)
        return None if sl is None else (sl, parts)

    def wait_discriminator_side(self):
        self.d_bucket.finish(self.world_size)

    def reduce(self):
        """apex-compatible entry point: reduce everything that currently has a gradient"""
        self.reduce_generator_side()
        self.reduce_discriminator_side()
