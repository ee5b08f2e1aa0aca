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
NO_REDUCER = os.environ.get('DIAG_NO_REDUCER', '0') != '0'      # no exchange, the SAME batch on every rank: pure process-to-process repeatability
if not NO_REDUCER:
    tm.reducer = GradReducer(tm, finetune=False, optimizer_G=opt_G, optimizer_D=opt_D, max_batch=8)
    args.num_gpus = world
else:
    args.num_gpus = 1
data, target = bench.synthetic_batch(args, 8, seed=123 + (0 if NO_REDUCER else rank))
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
    # element-level picture of the critic's (u, v) buffers vs rank 0
    for k, b in tm.discriminator.named_buffers():
        if b.dtype != torch.float32 or k.startswith('embed'):
            continue
        ref = b.detach().clone()
        dist.broadcast(ref, 0)
        d = (b - ref).abs()
        nz = (d > 0)
        st = torch.stack([d.max(), nz.sum().float(), (d / ref.abs().clamp_min(1e-30)).max()])
        idx = torch.nonzero(nz.reshape(-1))[:6, 0].tolist() if bool(nz.any()) else []
        allst = [None] * world
        dist.all_gather_object(allst, (st.tolist(), idx))
        if rank == 0 and any(a[0][1] > 0 for a in allst):
            a = max(allst, key=lambda q: q[0][1])
            print(f'[replicas]    {k} [{b.numel()}]: {int(a[0][1])} elements differ from rank 0, max |delta| {a[0][0]:.3e}, max relative {a[0][2]:.3e}, first indices {a[1]}', flush=True)
    # how far apart: max |delta| of the label embedding between rank 0 and the others (0 expected)
    w = tm.discriminator.embed.weight_orig.detach()
    ref = w.clone()
    dist.broadcast(ref, 0)
    d = (w - ref).abs()
    stat = torch.stack([d.max(), (d > 0).sum().float(), (d > 0).any(dim=1).sum().float()])
    dist.all_reduce(stat, op=dist.ReduceOp.MAX)
    if rank == 0 and float(stat[0]) > 0:
        print(f'[replicas]    label embedding vs rank 0: max |delta| {float(stat[0]):.3e}, {int(stat[1])} elements in {int(stat[2])} rows differ', flush=True)


def sn_probe(tag):
    """ONE extra power iteration of the critic's layers from the current (W, u, v) on every rank, compared across ranks, then undone"""
    D = tm.discriminator
    layers = D._conv_sn_layers()
    keep = [(l.weight_u.clone(), l.weight_v.clone()) for l in layers]
    was = [(int(l.weight_u.view(torch.int32).to(torch.int64).sum()), int(l.weight_v.view(torch.int32).to(torch.int64).sum()),
            int(l.weight_orig.detach().view(torch.int32).to(torch.int64).sum())) for l in layers]
    with torch.no_grad():
        D._sn_batch.update(True)
    torch.cuda.synchronize()
    now = [(int(l.weight_u.view(torch.int32).to(torch.int64).sum()), int(l.weight_v.view(torch.int32).to(torch.int64).sum())) for l in layers]
    for l, (u, v) in zip(layers, keep):
        l.weight_u.copy_(u); l.weight_v.copy_(v)
    allw, alln = [None] * world, [None] * world
    dist.all_gather_object(allw, was); dist.all_gather_object(alln, now)
    if rank == 0:
        names = {id(m): k for k, m in D.named_modules()}
        inp = [names[id(l)] for i, l in enumerate(layers) if any(a[i] != allw[0][i] for a in allw)]
        out = [names[id(l)] for i, l in enumerate(layers) if any(a[i] != alln[0][i] for a in alln)]
        print(f'[sn-probe] {tag}: inputs (u, v, W) differ between ranks for {len(inp)} layers {inp[:6]}; after ONE more power iteration outputs differ for {len(out)} layers {out[:6]}', flush=True)


# ---- trace of every power iteration of the critic's conv layers inside the eager steps: checksums of (W before, u, v after) + stream + set
from latent_pose_reenactment_amd import nn as lpnn  # noqa: E402
TRACE = []
CLONES = []
_orig_update = lpnn.SNBatch.update


SYNC = os.environ.get('DIAG_SYNC', 'trace')      # trace: synchronize + checksums around every update; before | after: only a device synchronize; none


def traced_update(self, training):
    mine = self is tm.discriminator.__dict__.get('_sn_batch') and not torch.cuda.is_current_stream_capturing()
    if mine and SYNC in ('trace', 'before'):
        torch.cuda.synchronize()
    if mine and SYNC == 'trace':
        wsum = [int(l.weight_orig.detach().view(torch.int32).to(torch.int64).sum()) for l in self.layers]
        usum0 = [int(l.weight_u.view(torch.int32).to(torch.int64).sum()) for l in self.layers]
    if mine and SYNC == 'clone':          # stream-ordered snapshots, no host synchronisation: (W, u, v) before and (u, v) after this power iteration
        before = [(l.weight_orig.detach().clone(), l.weight_u.clone(), l.weight_v.clone()) for l in self.layers]
    if mine and SYNC == 'exclusive':
        # the ranks take turns: while one rank runs its power iteration the other one's GPU queue is empty (does the OTHER process matter?)
        out = None
        for turn in range(world):
            torch.cuda.synchronize(); dist.barrier()
            if turn == rank:
                out = _orig_update(self, training)
                torch.cuda.synchronize()
        dist.barrier()
        return out
    out = _orig_update(self, training)
    if mine and SYNC == 'clone':
        CLONES.append((before, [(l.weight_u.clone(), l.weight_v.clone()) for l in self.layers], [s_[2].clone() for s_ in out]))
    if mine and SYNC in ('trace', 'after'):
        torch.cuda.synchronize()
    if mine and SYNC == 'trace':
        usum = [int(l.weight_u.view(torch.int32).to(torch.int64).sum()) for l in self.layers]
        TRACE.append((len(TRACE), int(torch.cuda.current_stream().cuda_stream), wsum, usum0, usum))
    return out


lpnn.SNBatch.update = traced_update

if os.environ.get('DIAG_TORCH_SN', '0') != '0':
    # the power iteration of the critic's conv layers with torch ops instead of lp_sn_power_iter (same semantics: spectral_norm.compute_weight)
    import torch.nn.functional as F_

    def torch_update(self, training):
        if self is not tm.discriminator.__dict__.get('_sn_batch'):
            return traced_update(self, training)
        states = []
        with torch.no_grad():
            for l in self.layers:
                w = l.weight_orig.detach().reshape(l.weight_orig.shape[0], -1)
                u, v = l.weight_u, l.weight_v
                if training:
                    v.copy_(F_.normalize(torch.mv(w.t(), u), dim=0, eps=l.eps))
                    u.copy_(F_.normalize(torch.mv(w, v), dim=0, eps=l.eps))
                sigma = torch.dot(u, torch.mv(w, v))
                states.append((u.clone(), v.clone(), torch.stack([sigma, 1.0 / sigma])))
        return states
    lpnn.SNBatch.update = torch_update


def clones_report(tag):
    if not CLONES:
        return
    torch.cuda.synchronize()
    cs = lambda t: int(t.reshape(-1).view(torch.int32).to(torch.int64).sum())
    # every rank checks ITS OWN power iterations against an fp64 torch evaluation from its own before-snapshots: err[i][j] = max |u_after - u_exact|
    errs = []
    for b, a, sg in CLONES:
        row = []
        for (w, u, v), (ua, va) in zip(b, a):
            w64 = w.reshape(w.shape[0], -1).double()
            v64 = torch.mv(w64.t(), u.double()); v64 = v64 / v64.norm().clamp_min(1e-4)
            u64 = torch.mv(w64, v64); u64 = u64 / u64.norm().clamp_min(1e-4)
            # ... and against two "wrong input" candidates: no iteration at all (u unchanged), or v taken as it was BEFORE (u = normalize(W v_before))
            u_stale = torch.mv(w64, v.double()); u_stale = u_stale / u_stale.norm().clamp_min(1e-4)
            row.append((float((ua.double() - u64).abs().max()), float((va.double() - v64).abs().max()), float((ua - u).abs().max()), float((va - v).abs().max()),
                        float((ua.double() - u_stale).abs().max())))
        errs.append(row)
    mine = [([(cs(w), cs(u), cs(v)) for w, u, v in b], [(cs(u), cs(v)) for u, v in a], [cs(x) for x in sg], e) for (b, a, sg), e in zip(CLONES, errs)]
    allm = [None] * world
    dist.all_gather_object(allm, mine)
    if rank == 0:
        names = [k for k, m in tm.discriminator.named_modules() if any(m is l for l in tm.discriminator._conv_sn_layers())]
        for i in range(min(len(m) for m in allm)):
            a, b = allm[0][i], allm[1][i]
            dw = [names[j] for j in range(len(names)) if a[0][j][0] != b[0][j][0]]
            dub = [names[j] for j in range(len(names)) if a[0][j][1] != b[0][j][1] or a[0][j][2] != b[0][j][2]]
            dua = [names[j] for j in range(len(names)) if a[1][j] != b[1][j]]
            dsg = [names[j] for j in range(len(names)) if a[2][j] != b[2][j]]
            print(f'[sn-clones] {tag} power iteration {i}: W differs {dw[:3]}; (u, v) before differ {dub[:3]}; (u, v) after differ {dua[:3]}; sigma differs {dsg[:3]}', flush=True)
            for j in range(len(names)):
                if a[1][j] != b[1][j]:
                    print(f'[sn-clones]     {names[j]}: |u - u_exact(fp64 from the rank\'s own snapshot)| rank 0 {a[3][j][0]:.2e}, rank 1 {b[3][j][0]:.2e};  |v - v_exact| rank 0 {a[3][j][1]:.2e}, rank 1 {b[3][j][1]:.2e}'
                          f' | rank 1: |u_after - u_before| {b[3][j][2]:.2e}, |v_after - v_before| {b[3][j][3]:.2e}, |u_after - normalize(W v_before)| {b[3][j][4]:.2e}'
                          f' | rank 0: {a[3][j][2]:.2e}, {a[3][j][3]:.2e}, {a[3][j][4]:.2e}', flush=True)
            worst = max(max(x[0] for x in a[3]), max(x[0] for x in b[3]))
            print(f'[sn-clones]     worst |u - u_exact| over all layers and both ranks in this iteration: {worst:.2e}', flush=True)
        # how do the vectors differ?  first differing layer of the last iteration on rank 0 vs rank 1 needs the tensors: gathered below
    # element-level picture for one differing layer (rank 1 sends its tensors of the first differing layer to rank 0)
    last = CLONES[-1][1]
    flat = torch.cat([torch.cat([u.reshape(-1), v.reshape(-1)]) for u, v in last])
    ref = flat.clone()
    dist.broadcast(ref, 0)
    d = (flat - ref).abs()
    st = torch.stack([d.max(), (d > 0).sum().float(), torch.tensor(float(flat.numel()), device=flat.device), (d / ref.abs().clamp_min(1e-30)).max()])
    dist.all_reduce(st, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(f'[sn-clones] {tag}: all (u, v) after the last power iteration vs rank 0: max |delta| {float(st[0]):.3e}, {int(st[1])} of {int(st[2])} elements differ, max relative {float(st[3]):.3e}', flush=True)
    del CLONES[:]


def trace_report(tag):
    allt = [None] * world
    dist.all_gather_object(allt, list(TRACE))
    del TRACE[:]
    if rank == 0:
        names = [k for k, m in tm.discriminator.named_modules() if any(m is l for l in tm.discriminator._conv_sn_layers())]
        n = min(len(t) for t in allt)
        print(f'[sn-trace] {tag}: {[len(t) for t in allt]} power iterations per rank', flush=True)
        for i in range(n):
            a, b = allt[0][i], allt[1][i]
            dw = [names[j] for j in range(len(a[2])) if a[2][j] != b[2][j]]
            du0 = [names[j] for j in range(len(a[3])) if a[3][j] != b[3][j]]
            du = [names[j] for j in range(len(a[4])) if a[4][j] != b[4][j]]
            print(f'[sn-trace]   iteration {i}: streams {a[1]:#x} / {b[1]:#x}; W differs before: {dw[:4]}; u differs before: {du0[:4]}; u differs after: {du[:4]}', flush=True)


def embed_probe(tag, n=3):
    """n power iterations of the LABEL EMBEDDING (98000 x 512) from the current state on every rank, compared across ranks after each, then undone"""
    D = tm.discriminator
    e = D.embed
    keep = (e.weight_u.clone(), e.weight_v.clone())
    res = []
    for _ in range(n):
        with torch.no_grad():
            st = D._embed_batch().update(True)
        torch.cuda.synchronize()
        res.append((int(e.weight_u.view(torch.int32).to(torch.int64).sum()), int(e.weight_v.view(torch.int32).to(torch.int64).sum()),
                    float(st[0][2][0]), int(st[0][2].view(torch.int32).to(torch.int64).sum())))
    e.weight_u.copy_(keep[0]); e.weight_v.copy_(keep[1])
    allr = [None] * world
    dist.all_gather_object(allr, res)
    if rank == 0:
        print(f'[embed-probe] {tag}: ' + '; '.join(f'iteration {i}: ' + ('same' if all(a[i] == allr[0][i] for a in allr) else f'DIFFERENT {[a[i] for a in allr]}') for i in range(n)), flush=True)


compare('after the start-up broadcast')
sn_probe('after the start-up broadcast')
embed_probe('after the start-up broadcast')
del TRACE[:]
if mode == 'graph':
    step = holycow.GraphedTrainStep(tm, opt_G, opt_D, args, data, target, warmup_steps=1)
    compare('after 1 eager warm-up step + capture')
else:
    def step():
        holycow.train_step(tm, data, target, opt_G, opt_D, args)
def phase_check(tag):
    """checksums of the critic's (u, v) buffers and weights across ranks at a point INSIDE a step"""
    torch.cuda.synchronize()
    D = tm.discriminator
    cur = [(k, int(b.detach().reshape(-1).view(torch.int32).to(torch.int64).sum())) for k, b in list(D.named_buffers()) + list(D.named_parameters()) if b.dtype == torch.float32]
    allc = [None] * world
    dist.all_gather_object(allc, cur)
    if rank == 0:
        bad = [cur[i][0] for i in range(len(cur)) if any(c[i][1] != allc[0][i][1] for c in allc)]
        print(f'[phase] {tag}: {len(bad)} critic tensors differ' + (': ' + ', '.join(bad[:6]) if bad else ''), flush=True)


def phased_step():
    """runners.holycow.train_step (data-parallel path with the two generator-side buckets), with a cross-rank check of the critic after every phase"""
    from latent_pose_reenactment_amd.nn import fused_grad_accumulation
    from latent_pose_reenactment_amd import streams as _streams
    reducer = tm.reducer
    tm.__dict__['_ebwd_cut'] = True
    try:
        all_data, losses_G, losses_D = tm(data, target)
    finally:
        tm.__dict__['_ebwd_cut'] = False
    phase_check('forward')
    loss_G = sum(v for v in losses_G.values()); loss_D = sum(v for v in losses_D.values())
    opt_G.zero_grad()
    with fused_grad_accumulation():
        loss_G.backward(retain_graph=True)
    _streams.join_all()
    phase_check('loss_G.backward')
    reducer.reduce_generator_side(async_op=True, part='generator')
    with fused_grad_accumulation():
        tm.embedder_backward()
    _streams.join_all()
    reducer.reduce_generator_side(async_op=True, part='embedder')
    opt_D.zero_grad()
    with fused_grad_accumulation():
        loss_D.backward()
    _streams.join_all()
    phase_check('loss_D.backward')
    reducer.wait_generator_side()
    opt_G.step()
    phase_check('optimizer_G.step')
    reducer.reduce_discriminator_side()
    phase_check('discriminator-side exchange')
    opt_D.step()
    phase_check('optimizer_D.step')
    tm.update_running_average(0.999)
    phase_check('EMA')


if os.environ.get('DIAG_PHASES', '0') != '0' and mode == 'eager':
    step = phased_step
for i in range(steps):
    step()
    torch.cuda.synchronize()
    trace_report(f'{mode} step {i + 1}')
    clones_report(f'{mode} step {i + 1}')
    compare(f'after {mode} step {i + 1}')
    sn_probe(f'after {mode} step {i + 1}')
    embed_probe(f'after {mode} step {i + 1}')
    del TRACE[:]
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
