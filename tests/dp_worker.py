"""Worker of tests/test_data_parallel_gpu.py: one rank of a `gloo` group whose ranks share the single GPU of the test box (RCCL
refuses two ranks on one device; the collectives' semantics are the same).  Builds the small meta-training modules with identical
seeds on every rank, takes its slice of the global batch, runs the REAL train step (runners.holycow.train_step or the re-cut
GraphedTrainStep) with parallel.GradReducer, and -- rank 0 -- saves the gradient arenas / updated weights."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))


ORDER = []


def small_args(world, rank, num_labels):
    return argparse.Namespace(image_size=32, num_channels=4, max_num_channels=16, embed_channels=8, pose_embedding_size=4, in_channels=3,
                              out_channels=3, num_labels=num_labels, dis_num_blocks=5, gen_padding='zero', norm_layer='in',
                              gen_constant_input_size=4, gen_num_residual_blocks=2, dis_padding='zero', device='cuda', optimizer='Adam',
                              lr_gen=5e-5, lr_dis=2e-4, beta1=0.0, finetune=False, num_gpus=world, world_size=world, rank=rank,
                              average_function='sum', dis_embed_weight=1e-2)


def build(a, seed=7):
    from generators.vector_pose_unsupervised_segmentation_noBottleneck import Wrapper as GW
    from embedders.unsupervised_pose_separate_embResNeXt_segmentation import Wrapper as EW
    from discriminators.no_landmarks import Wrapper as DW
    from criterions import adversarial, featmat, dis_embed
    from runners import holycow
    torch.manual_seed(seed)
    D, G, E = DW.get_net(a), GW.get_net(a), EW.get_net(a)
    with torch.no_grad():
        G.constant.constant.normal_()
    # mean-type losses only: dice is a ratio of batch sums (rank-local in the reference too), so it has no global-batch equivalent
    crits = [adversarial.Criterion('gan'), featmat.Criterion(10.0), dis_embed.Criterion(1e-2)]
    tm = holycow.TrainingModule(E, G, D, crits, [], {})
    opt_G = holycow.get_optimizer(tm.embedder, tm.generator, a)
    opt_D = DW.get_optimizer(tm.discriminator, a)
    tm.train()
    tm.embedder.eval()           # BatchNorm on running statistics, no dropout: every sample independent of its batch (and of the RNG)
    return tm, opt_G, opt_D, holycow


def global_batch(total, num_labels, seed=11):
    g = torch.Generator().manual_seed(seed)
    data = {'enc_rgbs': torch.rand(total, 2, 3, 32, 32, generator=g), 'pose_input_rgbs': torch.rand(total, 1, 3, 32, 32, generator=g),
            'target_rgbs': torch.rand(total, 1, 3, 32, 32, generator=g)}
    target = {'real_segm': torch.rand(total, 1, 1, 32, 32, generator=g).expand(total, 1, 3, 32, 32).contiguous(),
              'label': torch.randint(0, num_labels, (total,), generator=g)}
    return data, target


def run(world, rank, mode, total, num_labels, steps, out_path):
    # 32-px frames are outside the hand-written encoders' geometry (the product raises): this worker tests the data-parallel EXCHANGE, its
    # encoders run on the oracle's stock layers (test infrastructure) for the whole run
    from oracle import backbones_ref as BR
    with BR.stock_layers():
        _run(world, rank, mode, total, num_labels, steps, out_path)


def _run(world, rank, mode, total, num_labels, steps, out_path):
    a = small_args(world, rank, num_labels)
    tm, opt_G, opt_D, holycow = build(a)
    if world > 1:
        from latent_pose_reenactment_amd.parallel import GradReducer
        tm.reducer = GradReducer(tm, finetune=False, optimizer_G=opt_G, optimizer_D=opt_D)
    data, target = global_batch(total, num_labels)
    per = total // world
    sl = slice(rank * per, (rank + 1) * per)
    data = {k: v[sl].cuda() for k, v in data.items()}
    target = {k: v[sl].cuda() for k, v in target.items()}
    if mode == 'graph':
        # first step eager (lazy state -- optimizer moments, weight packs, MIOpen plans -- must exist before a capture: state created
        # INSIDE a capture would be re-initialised by every replay), then the captured step replayed for the rest
        holycow.train_step(tm, data, target, opt_G, opt_D, a)
        step = holycow.GraphedTrainStep(tm, opt_G, opt_D, a, data, target, warmup_steps=0)
        if world > 1:
            # order log of the re-cut step: graph replays and collectives as the host issues them (the test asserts that the generator-side
            # all-reduce is issued ASYNCHRONOUSLY before the discriminator-backward graph and waited for only before optimizer_G's graph)
            class _Logged:
                def __init__(self, name, g):
                    self.name, self.g = name, g

                def replay(self):
                    ORDER.append(self.name)
                    self.g.replay()
            for nm in ('g1', 'g1b', 'g2a', 'g2b', 'g3'):
                if hasattr(step, nm):
                    setattr(step, nm, _Logged(nm, getattr(step, nm)))
            orig_ar, orig_wait = dist.all_reduce, step.reducer.wait_generator_side
            arena_g = opt_G.ensure_flat(0)
            g_lo, g_hi, n_gen = arena_g.data_ptr(), arena_g.data_ptr() + arena_g.numel() * 4, step.reducer.n_gen

            def logged_all_reduce(t, *args, **kw):
                if g_lo <= t.data_ptr() < g_hi:          # a slice of the generator-side arena: the generator's bucket or the encoders'
                    tag = 'G-generator' if (t.data_ptr() == g_lo and t.numel() == n_gen) else 'G-embedder' if t.data_ptr() == g_lo + 4 * n_gen else 'G'
                else:
                    tag = 'D-side'
                ORDER.append(('all_reduce', tag, bool(kw.get('async_op', False))))
                return orig_ar(t, *args, **kw)

            def logged_wait():
                ORDER.append('wait_G')
                return orig_wait()
            dist.all_reduce = logged_all_reduce
            step.reducer.wait_generator_side = logged_wait
        for _ in range(steps - 1):
            del ORDER[:]
            step()
    else:
        for _ in range(steps):
            holycow.train_step(tm, data, target, opt_G, opt_D, a)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({'order': list(ORDER), 'gradG': opt_G.ensure_flat(0).cpu(), 'gradD': opt_D.ensure_flat(0).cpu(),
                    'G': {k: v.cpu() for k, v in tm.generator.state_dict().items()},
                    'D': {k: v.cpu() for k, v in tm.discriminator.state_dict().items()}}, out_path)


if __name__ == '__main__':
    mode, total, num_labels, steps, out_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group(backend='gloo', init_method='env://')
    run(world, rank, mode, total, num_labels, steps, out_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
