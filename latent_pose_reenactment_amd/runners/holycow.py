"""Training runner (reference API: runners/holycow.py:18-402): ``get_args``, ``get_optimizer``, ``TrainingModule``,
``run_epoch``.  The step ordering of run_epoch is part of parity:
  zero_grad(G) -> loss_G.backward(retain_graph) -> [all-reduce] -> step(G) -> zero_grad(D) -> loss_D.backward() ->
  [all-reduce] -> step(D) -> EMA(0.999 | 0.972 when fine-tuning).
The reference's apex ``Reducer`` is replaced by ``parallel.GradReducer`` (RCCL all-reduce over xGMI of exactly the
gradients each optimizer is about to consume).  TensorBoard / image-grid logging branches (holycow.py:266-400) are
observability and out of scope; scalar losses go to ``Meter`` as in the reference."""
import contextlib
import copy
import os
import itertools
import logging
import time

import torch
from torch import nn

from latent_pose_reenactment_amd import hipops as ops
from latent_pose_reenactment_amd import streams as _streams
from latent_pose_reenactment_amd.nn import fused_grad_accumulation
from latent_pose_reenactment_amd.utils import radam as _radam
from latent_pose_reenactment_amd.utils.utils import Meter, dict_to_device
from latent_pose_reenactment_amd.utils.tracing import rng

torch.optim.RAdam = _radam.RAdam
logger = logging.getLogger('runner')


def get_args(parser):
    parser.add('--iteration', type=int, default=0, help="Optional iteration number to start from")
    parser.add('--log_frequency_loss', type=int, default=1)
    parser.add('--log_frequency_images', type=int, default=100)
    parser.add('--log_frequency_fixed_images', type=int, default=2500)
    parser.add('--detailed_metrics', action='store_bool', default=True)
    parser.add('--num_visuals_per_img', default=2, type=int)
    parser.add('--fixed_val_ids', action='append', type=int, default=[50, 100, 200, 250, 300])
    parser.add('--batch_size_inference', default=5, type=int)
    return parser


def optimizer_class(name, device):
    """Adam | RAdam (train.py --optimizer).  On the GPU the fused multi-tensor HIP kernels are used (one launch per step,
    device-side step counter); on CPU the plain Python implementations with identical update rules."""
    if str(device).startswith('cuda'):
        from latent_pose_reenactment_amd.optim import FusedAdam, FusedRAdam
        return {'Adam': FusedAdam, 'RAdam': FusedRAdam}[name]
    return {'Adam': torch.optim.Adam, 'RAdam': _radam.RAdam}[name]


def get_optimizer(embedder, generator, args):
    params = list(generator.parameters())
    if not getattr(args, 'finetune', False):
        params += list(embedder.parameters())
    return optimizer_class(args.optimizer, args.device)(params, lr=args.lr_gen, betas=(args.beta1, 0.999), eps=1e-5)


class _Toggle:
    """attribute override usable both as a plain call and as a context manager (holycow.py:111-151)"""

    def __init__(self, obj, name, value):
        self.obj, self.name, self.old = obj, name, getattr(obj, name)
        setattr(obj, name, value)

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        setattr(self.obj, self.name, self.old)


class TrainingModule(nn.Module):
    def __init__(self, embedder, generator, discriminator, criterion_list, metric_list, running_averages={}):
        super().__init__()
        self.embedder, self.generator, self.discriminator = embedder, generator, discriminator
        self.criterion_list = nn.ModuleList(criterion_list)
        self.metric_list = nn.ModuleList(metric_list)
        self.compute_losses = True
        self.use_running_averages = False
        self.initialize_running_averages(running_averages)

    def initialize_running_averages(self, initial_values={}):
        """EMA copies of embedder and generator kept in a plain dict (not sub-modules: they are neither broadcast nor
        reduced nor optimised); ``None`` disables them -- holycow.py:65-97."""
        self.running_averages = {}
        if initial_values is not None:
            for name in ('embedder', 'generator'):
                avg = copy.deepcopy(getattr(self, name))
                if name in initial_values:
                    try:
                        avg.load_state_dict(initial_values[name])
                    except Exception:
                        logger.warning(f"running-average state of {name} does not fit the module; initialising by cloning")
                        avg.load_state_dict(getattr(self, name).state_dict())
                else:
                    logger.info(f"No initial value of weights' running averages provided for {name}. Initializing by cloning")
                self.running_averages[name] = avg
        for module in self.running_averages.values():
            module.eval()
            module.requires_grad_(False)

    def update_running_average(self, alpha=0.999, only=None, skip=()):
        """``only`` / ``skip``: names of the averaged modules to update / to leave out (round 6: the generator's average is updated beside the encoders'
        backward, the rest at the end of the step)"""
        with torch.no_grad():
            for name, avg in self.running_averages.items():
                if (only is not None and name not in only) or name in skip:
                    continue
                cur = getattr(self, name)
                first = next(iter(cur.parameters()), None)
                if first is not None and first.is_cuda:
                    from latent_pose_reenactment_amd.optim import FusedEMA
                    cache = self.__dict__.setdefault('_fused_ema', {})
                    if name not in cache or not cache[name].valid():
                        cache[name] = FusedEMA(cur, avg)
                    cache[name].update(alpha)       # two launches for the whole module (lp_mt_ema)
                    continue
                for p, p_avg in zip(cur.parameters(), avg.parameters()):
                    p_avg.mul_(alpha).add_(p * (1 - alpha))
                for b, b_avg in zip(cur.buffers(), avg.buffers()):
                    b_avg.copy_(b)

    def set_use_running_averages(self, use_running_averages=True):
        return _Toggle(self, 'use_running_averages', use_running_averages)

    def set_compute_losses(self, compute_losses=True):
        return _Toggle(self, 'compute_losses', compute_losses)

    def forward(self, data_dict, target_dict):
        if self.running_averages and self.use_running_averages:
            embedder, generator = self.running_averages['embedder'], self.running_averages['generator']
        else:
            embedder, generator = self.embedder, self.generator
        data_dict = copy.copy(data_dict)          # inputs only; modules add their outputs
        # the target-image halves of the VGG criterions depend on neither encoder nor generator -- they start now, each on a side stream
        # (the stream its criterion uses later where the criterions themselves run on side streams: meta-training; else joined before the call)
        from latent_pose_reenactment_amd import streams
        ft = bool(getattr(generator, 'finetuning', False))
        tgt = data_dict.get('target_rgbs')
        if tgt is None and isinstance(target_dict, dict):
            tgt = target_dict.get('target_rgbs')
        crit_stream = {}
        crit_side = self.compute_losses and streams.enabled(tgt, 'criterions', finetuning=ft)
        ahead = self.compute_losses and streams.enabled(tgt, 'targets', finetuning=ft)
        # LP_OVERLAP_TARGETS=2: start them only when the encoders are done, i.e. beside the GENERATOR's forward alone (its 4x4 .. 32x32 layers
        # are short launches that leave most of the chip idle)
        ahead_late = ahead and os.environ.get('LP_OVERLAP_TARGETS') == '2'

        def start_targets():
            both = {**data_dict, **target_dict}
            for i, criterion in enumerate(self.criterion_list):
                if i in crit_stream and hasattr(criterion, 'precompute_targets'):
                    with streams.branch(tgt.device, crit_stream[i]):
                        criterion.precompute_targets(both)
        if crit_side or ahead:
            for i, criterion in enumerate(self.criterion_list):
                if getattr(criterion, 'independent_branch', False):
                    crit_stream[i] = 1 + len(crit_stream)
            if ahead and not ahead_late:
                start_targets()
        # meta-training: what a training forward of G and D derives from the weights alone (spectral-norm power iterations, 16-bit weight
        # packs: ~40 short launches) is issued on a side stream beside the encoders' large kernels and joined before the generator runs
        prep = None
        if self.compute_losses and self.training and torch.is_grad_enabled() and streams.enabled(tgt, 'prepare', finetuning=ft) \
                and hasattr(generator, 'prepare_step') and hasattr(self.discriminator, 'prepare_step'):
            with streams.branch(tgt.device, 5) as prep:
                generator.prepare_step()
                self.discriminator.prepare_step()
        # In fine-tuning the optimizer holds generator parameters only (get_optimizer above, holycow.py:34-41), so the pose
        # encoder's weight gradients are never consumed: run it without autograd (bit-identical parameters afterwards).
        with torch.set_grad_enabled(torch.is_grad_enabled() and not getattr(generator, 'finetuning', False)), rng('forward.embedder'):
            embedder(data_dict)
        # ``_ebwd_cut`` (set by train_step / GraphedTrainStep, one GPU, meta-training): the autograd graph is CUT behind the embedder -- the
        # generator, discriminator and criterions see leaf copies of its outputs, loss_G.backward stops there, and ``embedder_backward()``
        # later continues into the encoders with the gradients those leaves collected.  Same arithmetic; it lets the encoders' backward
        # (bandwidth-bound kernels) run BESIDE loss_D.backward (the discriminator's convolutions) instead of in front of it.
        cut = None
        if self.__dict__.get('_ebwd_cut') and torch.is_grad_enabled() and self.compute_losses and not ft:
            cut = {}
            for k in ('embeds', 'embeds_elemwise', 'pose_embedding'):
                v = data_dict.get(k)
                if torch.is_tensor(v) and v.requires_grad:
                    leaf = v.detach().requires_grad_(True)
                    cut[k] = (v, leaf)
                    data_dict[k] = leaf
        self.__dict__['_ebwd_pending'] = cut
        if prep is not None:
            prep.join()
        if ahead_late:
            start_targets()
        if prep is not None and streams.enabled(tgt, 'real', finetuning=ft) and hasattr(self.discriminator, 'start_real_pass'):
            self.discriminator.start_real_pass({**data_dict, **target_dict})
        with rng('forward.generator'):
            generator(data_dict)
        data_dict.update(target_dict)
        # criterions that touch neither the discriminator nor each other (the two VGG stacks: ``independent_branch``) are issued on side
        # streams BEFORE the discriminator pass, so that their small-map layers fill the gaps of its launches (streams.py)
        early = {}
        fake = data_dict.get('fake_rgbs')
        if crit_side and streams.enabled(fake, 'criterions', finetuning=ft):
            for i, criterion in enumerate(self.criterion_list):
                if getattr(criterion, 'independent_branch', False):
                    with streams.branch(fake.device, crit_stream.get(i, 1 + len(early))) as b, rng('forward.criterion.' + type(criterion).__module__.split('.')[-1]):
                        early[i] = (b, criterion(data_dict))
        if self.compute_losses:
            with rng('forward.discriminator'):
                self.discriminator(data_dict)
        losses_G, losses_D = {}, {}
        for i, criterion in enumerate(self.criterion_list):
            try:
                if i in early:
                    b, out = early[i]
                    b.join(out)
                else:
                    if ahead and i in crit_stream:          # target features were computed on a side stream, the criterion itself runs here
                        torch.cuda.current_stream(tgt.device).wait_stream(streams.side_stream(tgt.device, crit_stream[i]))
                    with rng('forward.criterion.' + type(criterion).__module__.split('.')[-1]):
                        out = criterion(data_dict)
            except Exception:
                if self.compute_losses:
                    raise
                continue                          # visual/eval forwards lack targets: skip that loss
            if isinstance(out, tuple):
                if len(out) != 2:
                    raise TypeError(f'Unexpected number of outputs in criterion {type(criterion)}: expected 2, got {len(out)}')
                losses_G.update(out[0])
                losses_D.update(out[1])
            elif isinstance(out, dict):
                losses_G.update(out)
            else:
                raise TypeError(f'Unexpected type of {type(criterion)} output: expected dict or tuple of two dicts, got {type(out)}')
        return data_dict, losses_G, losses_D

    def embedder_backward(self):
        """second half of a cut backward pass (see ``forward``): from the gradients loss_G.backward left on the embedder-output leaves into
        the encoders; a no-op when the last forward was not cut"""
        cut = self.__dict__.pop('_ebwd_pending', None)
        if not cut:
            return
        outs = [v for v, leaf in cut.values() if leaf.grad is not None]
        grads = [leaf.grad for v, leaf in cut.values() if leaf.grad is not None]
        if outs:
            torch.autograd.backward(outs, grads)

    def compute_metrics(self, data_dict):
        meter = Meter()
        for metric in self.metric_list:
            values, counts = metric(data_dict)
            for k, v in values.items():
                meter.add(k, v, counts[k])
        return meter


def train_step(training_module, data_dict, target_dict, optimizer_G, optimizer_D, args):
    """one iteration of the hot loop (holycow.py:230-257); returns (all_data_dict, losses_G, losses_D).

    Data parallel (1 < num_gpus <= 8): the generator-side all-reduce is issued asynchronously right after loss_G.backward and
    overlaps zero_grad(D) + loss_D.backward; optimizer_G.step waits for it.  Moving optimizer_G.step behind loss_D.backward is
    result-identical: loss_D depends on fake.detach() and on tensors the discriminator saved in the forward pass, never on the
    generator's parameters or gradients (the reference reduces, steps G, then runs the D backward, holycow.py:239-250)."""
    reducer = getattr(training_module, 'reducer', None)
    multi = 1 < args.num_gpus <= 8 and reducer is not None
    ebwd = _ebwd_enabled(training_module, args, multi)
    split = multi and _dp_split_enabled(training_module, args)
    training_module.__dict__['_ebwd_cut'] = ebwd or split
    try:
        all_data, losses_G, losses_D = training_module(data_dict, target_dict)
    finally:
        training_module.__dict__['_ebwd_cut'] = False          # (only this step's forward is cut: other callers get the plain graph)
    loss_G = sum(v for v in losses_G.values() if isinstance(v, torch.Tensor))
    loss_D = sum(v for v in losses_D.values() if isinstance(v, torch.Tensor))
    optimizer_G.zero_grad()
    gwg = ebwd and bool(losses_D) and _gwgrad_enabled(training_module)
    with fused_grad_accumulation(), rng('backward.loss_G'), (ops.wgrad_defer() if gwg else contextlib.nullcontext()):
        loss_G.backward(retain_graph=True)
    _streams.join_all()
    if ebwd and losses_D:
        # one GPU, meta-training: the encoders' half of the generator-side backward runs beside the discriminator-side backward
        # (``TrainingModule.forward`` cut the graph behind the embedder).  loss_D.backward is CALLED from a side stream: autograd orders
        # every node behind the stream the root gradient lives on, which must not be the stream the encoders' backward is queued on.
        dev = next(training_module.generator.parameters()).device
        early = _early_updates(training_module, optimizer_G)
        alpha = 0.972 if args.finetune else 0.999
        with _streams.branch(dev, 8) as b:
            optimizer_D.zero_grad()
            with fused_grad_accumulation(), rng('backward.loss_D'):
                # (round 6) the generator's weight gradients, recorded during loss_G.backward (hipops.wgrad_defer), are issued HERE: matrix-bound
                # launches beside the encoders' bandwidth-bound backward instead of inside the chain that the encoders' backward waits for
                held = ops.wgrad_flush() if gwg else None
                loss_D.backward()
            _streams.join_all()
            if early:
                # (round 6) everything that does not wait for the encoders' backward runs HERE, beside it: the critic's update, the generator's
                # slice of optimizer_G (its gradients are final since loss_G.backward) and the generator's running average -- none of them reads
                # or writes anything the encoders' backward touches.  What is left behind the join: the encoders' slice and their average.
                with rng('optimizer_D.step'):
                    optimizer_D.step()
                with rng('optimizer_G.step.generator'):
                    optimizer_G.step(part='generator')
                with rng('ema.generator'):
                    training_module.update_running_average(alpha, only=('generator',))
        with fused_grad_accumulation(), rng('backward.embedder'):
            training_module.embedder_backward()
        _streams.join_all()
        b.join()
        held = None          # (the deferred launches' operand planes: alive until their stream was joined)
        with rng('optimizer_G.step'):
            optimizer_G.step()
        if not early:
            with rng('optimizer_D.step'):
                optimizer_D.step()
        with rng('ema'):
            training_module.update_running_average(alpha, skip=('generator',) if early else ())
        return all_data, losses_G, losses_D
    if split:
        # data parallel, meta-training: the backward pass was cut behind the embedder -- the generator's gradients (the first 150 MB of the
        # generator-side arena) are final NOW and go out as their own bucket; the encoders' backward (>= 10 ms of kernels) runs while that
        # all-reduce is on the links, then the encoders' bucket follows; both hide behind zero_grad(D) + loss_D.backward
        with rng('all_reduce.generator_bucket.issue'):
            reducer.reduce_generator_side(async_op=True, part='generator')
        with fused_grad_accumulation(), rng('backward.embedder'):
            training_module.embedder_backward()
        _streams.join_all()
        with rng('all_reduce.encoder_bucket.issue'):
            reducer.reduce_generator_side(async_op=True, part='embedder')
    else:
        with fused_grad_accumulation(), rng('backward.embedder'):
            training_module.embedder_backward()          # (cut without a discriminator-side loss: finish the backward pass here)
        if multi:
            with rng('all_reduce.generator_side.issue'):
                reducer.reduce_generator_side(async_op=True)
        else:
            with rng('optimizer_G.step'):
                optimizer_G.step()
    if losses_D:
        optimizer_D.zero_grad()
        with fused_grad_accumulation(), rng('backward.loss_D'):
            loss_D.backward()
        _streams.join_all()
    if multi:
        with rng('all_reduce.generator_side.wait'):
            reducer.wait_generator_side()
        with rng('optimizer_G.step'):
            optimizer_G.step()
    if losses_D:
        if multi:
            with rng('all_reduce.discriminator_side'):
                reducer.reduce_discriminator_side()
        with rng('optimizer_D.step'):
            optimizer_D.step()
    with rng('ema'):
        training_module.update_running_average(0.972 if args.finetune else 0.999)
    return all_data, losses_G, losses_D


def _early_updates(training_module, optimizer_G):
    """one GPU, encoders' backward beside loss_D.backward: may optimizer_D.step, the generator's slice of optimizer_G and the generator's running
    average run beside the encoders' backward too?  Needs the fused optimizers (``set_partitions``); LP_OVERLAP_EARLY=0 keeps them behind the join."""
    if os.environ.get('LP_OVERLAP_EARLY', '1') == '0' or not hasattr(optimizer_G, 'set_partitions'):
        return False
    gen = list(training_module.generator.parameters())
    if optimizer_G.__dict__.get('_lp_gen_key') != tuple(id(p) for p in gen):
        optimizer_G.set_partitions({'generator': gen})
        optimizer_G.__dict__['_lp_gen_key'] = tuple(id(p) for p in gen)
    return True


def _gwgrad_enabled(training_module):
    """the generator's weight gradients deferred to the critic-backward stream (with 'ebwd'; LP_OVERLAP_GWGRAD)"""
    p = next(iter(training_module.generator.parameters()), None)
    return p is not None and _streams.enabled(p, 'gwgrad', finetuning=False)


def _ebwd_enabled(training_module, args, multi):
    """encoders' backward beside loss_D.backward: one GPU (the data-parallel step keeps the generator-side gradients in ONE backward pass so
    that their all-reduce can start early), meta-training (fine-tuning trains no encoder), training mode, LP_OVERLAP_EBWD != 0"""
    if multi or getattr(args, 'finetune', False) or not training_module.training:
        return False
    p = next(training_module.generator.parameters(), None)
    return p is not None and _streams.enabled(p, 'ebwd', finetuning=False)


def _dp_split_enabled(training_module, args):
    """data-parallel step with the generator-side exchange in two buckets (see ``train_step``): meta-training (fine-tuning trains no
    encoder, and the reference refuses multi-GPU fine-tuning: train.py:120-126), training mode, LP_DP_SPLIT != 0"""
    return not getattr(args, 'finetune', False) and training_module.training and os.environ.get('LP_DP_SPLIT', '1') != '0'


GRAPH_WARMUP_ITERATIONS = 3      # eager iterations before the step is captured (lazy state: optimizer moments, packs, MIOpen plans)


def _graphed_step(training_module, data_dict, target_dict, optimizer_G, optimizer_D, args):
    """``--hip_graph``: the first GRAPH_WARMUP_ITERATIONS iterations run eagerly (on the capture stream), then the step is captured
    once (capturing executes nothing) and every further iteration copies its batch into the static input buffers and replays the
    graphs -- the same sequence of optimizer steps as the eager loop.  A change of batch shape or of the optimizers re-captures."""
    st = training_module.__dict__.setdefault('_graph_state', {'eager_done': 0, 'step': None, 'key': None, 'stream': torch.cuda.Stream()})
    key = (id(optimizer_G), id(optimizer_D), training_module.training,
           tuple((k, tuple(v.shape)) for k, v in sorted(data_dict.items()) if torch.is_tensor(v)),
           tuple((k, tuple(v.shape)) for k, v in sorted(target_dict.items()) if torch.is_tensor(v)))
    if st['key'] != key:
        st.update(eager_done=0, step=None, key=key)
    if st['step'] is None and st['eager_done'] < GRAPH_WARMUP_ITERATIONS:
        st['stream'].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st['stream']):
            out = train_step(training_module, data_dict, target_dict, optimizer_G, optimizer_D, args)
        torch.cuda.current_stream().wait_stream(st['stream'])
        st['eager_done'] += 1
        return out
    if st['step'] is None:
        st['step'] = GraphedTrainStep(training_module, optimizer_G, optimizer_D, args, data_dict, target_dict, warmup_steps=0,
                                      stream=st['stream'])
    g = st['step']
    g.load_batch(data_dict, target_dict)
    g()
    return g.all_data, g.losses_G, g.losses_D


def run_epoch(dataloader, training_module, optimizer_G, optimizer_D, epoch, args, phase, writer=None, saver=None):
    meter = Meter()
    if phase == 'train':
        optimizer_G.zero_grad()
        if optimizer_D:
            optimizer_D.zero_grad()
    use_graph = phase == 'train' and getattr(args, 'hip_graph', False) and str(args.device).startswith('cuda')
    if phase == 'train' and str(args.device).startswith('cuda') and getattr(args, 'prefetch_to_device', True):
        # pinned, double-buffered H2D on a side stream, one batch ahead of the computing step (dataloaders/prefetch.py).  LIFETIME:
        # the batches handed out are views of THREE rotating device slots -- a tensor of data_dict / target_dict (or of all_data, which
        # aliases the inputs) that is kept for more than two further iterations (visualisation, saver hooks, debugging) is overwritten by a
        # later copy: clone what must outlive the iteration, or pass --prefetch_to_device False.
        from latent_pose_reenactment_amd.dataloaders.prefetch import DevicePrefetcher
        dataloader = DevicePrefetcher(dataloader, args.device)
    log_every = max(1, int(getattr(args, 'log_frequency_loss', 1)))
    end = time.time()
    for it, (data_dict, target_dict) in enumerate(dataloader):
        meter.add('Data_time', time.time() - end)
        dict_to_device(data_dict, args.device)
        dict_to_device(target_dict, args.device)
        if use_graph:
            all_data, losses_G, losses_D = _graphed_step(training_module, data_dict, target_dict, optimizer_G, optimizer_D, args)
        elif phase == 'train':
            all_data, losses_G, losses_D = train_step(training_module, data_dict, target_dict, optimizer_G, optimizer_D, args)
        else:
            all_data, losses_G, losses_D = training_module(data_dict, target_dict)
            if saver is not None:
                saver.save(epoch=epoch, data=all_data)
        # loss read-back = device->host sync, every iteration in the reference (holycow.py:260-262); with graph replay it is the
        # only sync of the loop, so it honours --log_frequency_loss there
        if args.detailed_metrics and (not use_graph or it % log_every == 0):
            for name, value in itertools.chain(losses_G.items(), losses_D.items()):
                meter.add(f'Loss_{name}', float(value))
        if phase == 'train':
            # The reference advances args.iteration inside its TensorBoard branch, i.e. on the logging rank only (holycow.py:319,
            # 389-390) -- the rank that also names the checkpoints model_{iteration:08}.pth.  Logging is out of scope here, so the
            # counter advances on every rank and every training iteration: same checkpoint names as a logging reference run.
            args.iteration += 1
        meter.add('Batch_time', time.time() - end)
        end = time.time()
    return meter


class GraphedTrainStep:
    """The training iteration of ``train_step`` captured into hipGraphs (MI355X: thousands of short launches per step make the
    eager loop host-bound; replaying graphs removes the per-launch CPU cost -- "HIP graphs instead of a tracing compiler").

    The graphs share one memory pool and are replayed in order; the data-parallel all-reduces run eagerly BETWEEN them
    (collectives are never captured).  One GPU:
        g1: forward (E, G, D x3, criterions) + zero_grad(G) + loss_G.backward
        g2: [optimizer_G.step + EMA on a side stream] || zero_grad(D) + loss_D.backward
        g3: optimizer_D.step
    Data parallel (the step is re-cut so that the generator-side exchange hides behind the encoders' and the discriminator's backward):
        g1:  as above, the backward pass stopping at the embedder's outputs (meta-training)
        --   all-reduce of the GENERATOR's slice of the generator-side gradient arena, ASYNCHRONOUS on RCCL's stream
        g1b: the encoders' backward                            (runs concurrently with that all-reduce)
        --   all-reduce of the ENCODERS' slice, asynchronous
        g2a: zero_grad(D) + loss_D.backward                    (runs concurrently with both)
        --   wait for the all-reduces
        g2b: optimizer_G.step
        --   discriminator-side exchange (arena all-reduce + row-sparse label-embedding exchange)
        g3:  optimizer_D.step + EMA
    Same arithmetic as the eager step (see ``train_step``).  Inputs are static buffers: ``load_batch`` copies a new batch into
    them.  Needs the fused (device-step-counter) optimizers; loss values live in ``losses_G`` / ``losses_D``."""

    def __init__(self, training_module, optimizer_G, optimizer_D, args, data_dict, target_dict, warmup_steps=3, stream=None):
        """``warmup_steps`` eager steps on the given batch precede the capture (they ARE optimizer steps: callers that must keep
        the eager trajectory pass 0 and warm up with their own real iterations on ``stream``, see ``_graphed_step``)."""
        self.tm, self.opt_G, self.opt_D, self.args = training_module, optimizer_G, optimizer_D, args
        self.data = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data_dict.items()}
        self.target = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in target_dict.items()}
        self.reducer = getattr(training_module, 'reducer', None) if 1 < args.num_gpus <= 8 else None
        self.alpha = 0.972 if args.finetune else 0.999
        side = stream if stream is not None else torch.cuda.Stream()
        self.stream = side
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                # eager warm-up: lazy state (optimizer moments, packs, MIOpen plans)
            for _ in range(warmup_steps):
                train_step(self.tm, self.data, self.target, self.opt_G, self.opt_D, args)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        G = torch.cuda.CUDAGraph
        kw = dict(stream=side, capture_error_mode='thread_local')
        # capture on the stream the warm-up ran on: the AccumulateGrad nodes autograd keeps per parameter stay on one stream
        # thread_local error mode: CUDA calls of OTHER threads (the RCCL watchdog polling its events) must not invalidate the capture
        self.ebwd = _ebwd_enabled(self.tm, args, self.reducer is not None)
        self.split = self.reducer is not None and _dp_split_enabled(self.tm, args)
        self.tm.__dict__['_ebwd_cut'] = self.ebwd or self.split
        self.g1 = G()
        with torch.cuda.graph(self.g1, **kw):
            try:
                self.all_data, self.losses_G, self.losses_D = self.tm(self.data, self.target)
            finally:
                self.tm.__dict__['_ebwd_cut'] = False
            loss_G = sum(v for v in self.losses_G.values() if isinstance(v, torch.Tensor))
            loss_D = sum(v for v in self.losses_D.values() if isinstance(v, torch.Tensor))
            self.opt_G.zero_grad()
            self.gwg = self.reducer is None and self.ebwd and bool(self.losses_D) and _gwgrad_enabled(self.tm)
            with fused_grad_accumulation(), (ops.wgrad_defer() if self.gwg else contextlib.nullcontext()):
                loss_G.backward(retain_graph=True)
            _streams.join_all()
        pool = self.g1.pool()
        from latent_pose_reenactment_amd import streams
        first = next(iter(self.tm.generator.parameters()))
        # one GPU: optimizer_G.step and the EMA of embedder + generator touch nothing the discriminator backward reads or writes
        # (train_step's docstring), so they run on a side stream beside it; g3 is then optimizer_D.step alone
        self.ema_in_g2 = self.reducer is None and streams.enabled(first, 'optimizer', finetuning=bool(getattr(args, 'finetune', False)))
        self.early = False
        if self.reducer is None and self.ebwd and self.losses_D:
            # g2: the encoders' backward (main path) beside zero_grad(D) + loss_D.backward (side path, called from its own stream: see
            # train_step), then optimizer_G.step
            self.ema_in_g2 = False
            self.early = _early_updates(self.tm, self.opt_G)          # (see train_step)
            self.g2 = G()
            with torch.cuda.graph(self.g2, pool=pool, **kw):
                with streams.branch(first.device, 8) as b:
                    self.opt_D.zero_grad()
                    with fused_grad_accumulation():
                        self._held = ops.wgrad_flush() if self.gwg else None          # (see train_step; the operand planes live as long as the graphs)
                        loss_D.backward()
                    _streams.join_all()
                    if self.early:
                        self.opt_D.step()
                        self.opt_G.step(part='generator')
                        self.tm.update_running_average(self.alpha, only=('generator',))
                with fused_grad_accumulation():
                    self.tm.embedder_backward()
                _streams.join_all()
                b.join()
                self.opt_G.step()
                if self.early:
                    self.tm.update_running_average(self.alpha, skip=('generator',))
        elif self.reducer is None:
            self.g2 = G()
            with torch.cuda.graph(self.g2, pool=pool, **kw):
                with fused_grad_accumulation():
                    self.tm.embedder_backward()          # (a cut forward without a discriminator-side loss: finish the backward pass; else a no-op)
                if self.ema_in_g2:
                    with streams.branch(first.device, 3) as b:
                        self.opt_G.step()
                        self.tm.update_running_average(self.alpha)
                else:
                    self.opt_G.step()
                self.opt_D.zero_grad()
                with fused_grad_accumulation():
                    loss_D.backward()
                _streams.join_all()
                if self.ema_in_g2:
                    b.join()
        else:
            if self.split:
                self.reducer.reduce_generator_side(part='generator')
                self.g1b = G()
                with torch.cuda.graph(self.g1b, pool=pool, **kw):
                    with fused_grad_accumulation():
                        self.tm.embedder_backward()
                    _streams.join_all()
                self.reducer.reduce_generator_side(part='embedder')
            else:
                self.reducer.reduce_generator_side()
            self.g2a, self.g2b = G(), G()
            with torch.cuda.graph(self.g2a, pool=pool, **kw):
                self.opt_D.zero_grad()
                with fused_grad_accumulation():
                    loss_D.backward()
                _streams.join_all()
            with torch.cuda.graph(self.g2b, pool=pool, **kw):
                self.opt_G.step()
            self.reducer.reduce_discriminator_side()
        self.g3 = None
        if not getattr(self, 'early', False):
            self.g3 = G()
            with torch.cuda.graph(self.g3, pool=pool, **kw):
                self.opt_D.step()
                if not self.ema_in_g2:
                    self.tm.update_running_average(self.alpha)
        del loss_G, loss_D
        torch.cuda.synchronize()

    def load_batch(self, data_dict, target_dict):
        for dst, src in ((self.data, data_dict), (self.target, target_dict)):
            for k, v in src.items():
                if torch.is_tensor(v):
                    dst[k].copy_(v, non_blocking=True)

    def __call__(self):
        from latent_pose_reenactment_amd.optim import WEIGHTS_GENERATION
        WEIGHTS_GENERATION[0] += 1          # replays update weights without touching any Python-side counter
        self.g1.replay()
        if self.reducer is None:
            self.g2.replay()
        else:
            if self.split:
                self.reducer.reduce_generator_side(async_op=True, part='generator')
                self.g1b.replay()
                self.reducer.reduce_generator_side(async_op=True, part='embedder')
            else:
                self.reducer.reduce_generator_side(async_op=True)
            self.g2a.replay()
            self.reducer.wait_generator_side()
            self.g2b.replay()
            self.reducer.reduce_discriminator_side()
        if self.g3 is not None:
            self.g3.replay()
