"""CPU tests of the host-side mirror of the reference: config merge order, store_bool flags, plugin loader, checkpoint
format round trip, Meter, embedder backbones' torchvision-compatible keys, and the RCCL reducer on a 2-rank gloo group."""
import argparse
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'latent_pose_reenactment_amd')
sys.path.insert(0, PKG)


def test_store_bool_and_config_resolution_order(tmp_path, monkeypatch):
    import importlib
    train = importlib.import_module('train')
    monkeypatch.chdir(PKG)
    monkeypatch.setattr(sys, 'argv', ['train.py', '--config_name=finetuning-base', '--lr_gen', '1e-3', '--no-logging',
                                      '--dataloader', 'synthetic_voxceleb2', '--generator', 'vector_pose_unsupervised_segmentation_noBottleneck',
                                      '--embedder', 'unsupervised_pose_separate_embResNeXt_segmentation', '--discriminator', 'no_landmarks',
                                      '--runner', 'holycow'])
    from utils.utils import get_args_and_modules
    args, default_args, m, ckpt = get_args_and_modules(train.build_parser(), use_checkpoint_args=True)
    assert args.finetune is True and args.optimizer == 'RAdam'          # from the yaml
    assert args.lr_gen == 1e-3                                           # command line beats yaml (5e-4)
    assert args.lr_dis == 8e-4 and args.logging is False                 # yaml beats plugin default; --no-logging
    assert args.criterions.replace(' ', '') == 'adversarial,featmat,idt_embed,perceptual,dice'
    assert [c.__module__ for c in m['criterion_list']] == ['criterions.adversarial', 'criterions.featmat', 'criterions.idt_embed',
                                                            'criterions.perceptual', 'criterions.dice']
    assert default_args.lr_gen == 5e-4 and ckpt is None            # parse_args([]) after set_defaults(yaml), as in the reference
    assert args.gen_num_residual_blocks == 2 and args.dis_num_blocks == 7 and args.num_labels == 98000


def test_meter_nan_handling():
    from utils.utils import Meter
    m = Meter()
    m.add('a', 2.0); m.add('a', float('nan')); m.add('a', 4.0, 3)
    assert m.get_average('a') == pytest.approx(14 / 4) and m.get_num_measurements('a') == 4
    other = Meter(); other.add('a', 1.0)
    m += other
    assert m.get_last('a') == 1.0


def test_backbone_keys_follow_torchvision_layout():
    from embedders.backbones import mobilenet_v2, resnext50_32x4d
    mb = mobilenet_v2(num_classes=256)
    keys = list(mb.state_dict().keys())
    assert keys[0] == 'features.0.0.weight' and 'features.1.conv.0.0.weight' in keys and 'features.2.conv.1.0.weight' in keys
    assert 'features.18.1.running_var' in keys and keys[-2:] == ['classifier.1.weight', 'classifier.1.bias']
    assert sum(p.numel() for p in mb.parameters()) == 2551808
    rx = resnext50_32x4d(num_classes=512)
    rk = rx.state_dict()
    assert rk['layer1.0.conv2.weight'].shape == (128, 4, 3, 3) and 'layer4.2.bn3.weight' in rk and 'layer1.0.downsample.0.weight' in rk
    assert sum(p.numel() for p in rx.parameters()) == 24028992


def test_checkpoint_round_trip_and_finetune_structure(tmp_path):
    """save_model writes the reference's dict layout; load_model_from_checkpoint rebuilds modules incl. the finetune switch"""
    from utils import utils
    from runners import holycow
    from generators.vector_pose_unsupervised_segmentation_noBottleneck import Wrapper as GW
    from embedders.unsupervised_pose_separate_embResNeXt_segmentation import Wrapper as EW
    from discriminators.no_landmarks import Wrapper as DW
    a = argparse.Namespace(image_size=16, num_channels=4, max_num_channels=8, embed_channels=8, pose_embedding_size=4, in_channels=3,
                           out_channels=3, gen_padding='zero', norm_layer='in', gen_constant_input_size=4, gen_num_residual_blocks=1,
                           dis_padding='zero', dis_num_blocks=3, num_labels=5, average_function='sum', device='cpu', optimizer='Adam',
                           lr_gen=1e-4, lr_dis=1e-4, beta1=0.0, finetune=False, rank=0, num_gpus=1, iteration=7,
                           experiment_dir=str(tmp_path), generator='vector_pose_unsupervised_segmentation_noBottleneck',
                           embedder='unsupervised_pose_separate_embResNeXt_segmentation', discriminator='no_landmarks', runner='holycow')
    E, G, D = EW.get_net(a), GW.get_net(a), DW.get_net(a)
    tm = holycow.TrainingModule(E, G, D, [], [], {})
    oG, oD = holycow.get_optimizer(E, G, a), DW.get_optimizer(D, a)
    path = utils.save_model(tm, oG, oD, a)
    assert os.path.basename(path) == 'model_00000007.pth'
    ck = utils.torch_load(path)
    assert set(ck) == {'embedder', 'generator', 'discriminator', 'optimizer_G', 'optimizer_D', 'running_averages', 'args'}
    assert set(ck['running_averages']) == {'embedder', 'generator'}
    a2 = argparse.Namespace(**vars(a)); a2.finetune = True
    E2, G2, D2, ra, saved, o1, o2 = utils.load_model_from_checkpoint(ck, a2)
    assert G2.finetuning and D2.finetuning and 'identity_embedding' in G2.state_dict() and D2.embed.weight_orig.shape == (1, 8)
    assert torch.equal(G2.constant.constant, G.constant.constant)


REDUCER_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'latent_pose_reenactment_amd'))
from latent_pose_reenactment_amd.parallel import GradReducer
rank = int(os.environ['RANK'])
dist.init_process_group('gloo', init_method='env://')
torch.manual_seed(0)
class TM(torch.nn.Module):
    def __init__(s):
        super().__init__()
        s.embedder = torch.nn.Linear(3, 2); s.generator = torch.nn.Linear(4, 3); s.discriminator = torch.nn.Linear(5, 1)
tm = TM()
if rank >= 1:
    for p in tm.parameters(): p.data.add_(float(rank))            # will be overwritten by the rank-0 broadcast
red = GradReducer(tm, finetune=False)
for p in tm.parameters(): p.grad = torch.full_like(p, float(rank + 1))
red.reduce_generator_side(async_op=True); red.wait_generator_side()
W = dist.get_world_size(); mean = (W + 1) / 2
ok = all(torch.allclose(p.grad, torch.full_like(p, mean)) for p in list(tm.generator.parameters()) + list(tm.embedder.parameters()))
ok &= all(torch.allclose(p.grad, torch.full_like(p, float(rank + 1))) for p in tm.discriminator.parameters())   # untouched so far
red.reduce_discriminator_side()
ok &= all(torch.allclose(p.grad, torch.full_like(p, mean)) for p in tm.discriminator.parameters())
# arena path: a (stand-in) fused optimizer owning one flat gradient buffer per side is reduced in place
class Arena:
    def __init__(s, params):
        s.params = list(params); s.param_groups = [{'params': s.params}]
        s.flat = torch.zeros(sum(p.numel() for p in s.params)); off = 0
        for p in s.params:
            p.grad = s.flat[off:off + p.numel()].view_as(p); off += p.numel()
    def ensure_flat(s, gi=0):
        return s.flat
oG, oD = Arena(list(tm.generator.parameters()) + list(tm.embedder.parameters())), Arena(tm.discriminator.parameters())
red2 = GradReducer(tm, finetune=False, broadcast=False, optimizer_G=oG, optimizer_D=oD)
oG.flat.fill_(float(rank + 1)); oD.flat.fill_(float(10 * (rank + 1)))
red2.reduce_generator_side(async_op=True); red2.wait_generator_side(); red2.reduce_discriminator_side()
ok &= bool(torch.allclose(oG.flat, torch.full_like(oG.flat, mean))) and bool(torch.allclose(oD.flat, torch.full_like(oD.flat, 10 * mean)))
ok &= all(torch.allclose(p.grad, torch.full_like(p, mean)) for p in tm.generator.parameters())
# generator-side exchange in TWO buckets (round 5): the generator's gradients first, the encoders' when their backward is done; one wait
tm3 = TM(); red4 = GradReducer(tm3, finetune=False)
for p in tm3.parameters(): p.grad = torch.full_like(p, float(rank + 1))
red4.reduce_generator_side(async_op=True, part='generator')
red4.reduce_generator_side(async_op=True, part='embedder'); red4.wait_generator_side()
ok &= all(torch.allclose(p.grad, torch.full_like(p, mean)) for p in list(tm3.generator.parameters()) + list(tm3.embedder.parameters()))
ok &= all(torch.allclose(p.grad, torch.full_like(p, float(rank + 1))) for p in tm3.discriminator.parameters())
oG3, oD3 = Arena(list(tm3.generator.parameters()) + list(tm3.embedder.parameters())), Arena(tm3.discriminator.parameters())
red5 = GradReducer(tm3, finetune=False, broadcast=False, optimizer_G=oG3, optimizer_D=oD3)
ok &= red5.n_gen == sum(p.numel() for p in tm3.generator.parameters()) and red5.n_gen + red5.n_emb == oG3.flat.numel()
oG3.flat.fill_(float(rank + 1))
red5.reduce_generator_side(async_op=False, part='generator')        # only the generator's slice of the arena is exchanged
ok &= bool(torch.allclose(oG3.flat[:red5.n_gen], torch.full((red5.n_gen,), mean))) and bool(torch.allclose(oG3.flat[red5.n_gen:], torch.full((red5.n_emb,), float(rank + 1))))
oG3.flat.fill_(float(rank + 1))
red5.reduce_generator_side(async_op=True, part='generator'); red5.reduce_generator_side(async_op=True, part='embedder'); red5.wait_generator_side()
ok &= bool(torch.allclose(oG3.flat, torch.full_like(oG3.flat, mean)))
w = [p.detach().clone() for p in tm.parameters()]
gathered = [None] * W; dist.all_gather_object(gathered, [t.tolist() for t in w])
ok &= all(g_ == gathered[0] for g_ in gathered)                            # parameters identical after the broadcast
# row-sparse label-embedding exchange (meta-training): every rank publishes (labels, B gradient rows, rank-1 coefficient); the rebuilt
# gradient must equal the average of the ranks' dense gradients  rows-scatter - coef * u v^T  -- also for labels above 2^24 (int64 exchange)
class SNW(torch.nn.Module):
    def __init__(s, n, e):
        super().__init__()
        s.weight_orig = torch.nn.Parameter(torch.randn(n, e)); s.register_buffer('weight_u', torch.randn(n)); s.register_buffer('weight_v', torch.randn(e))
class Dis(torch.nn.Module):
    def __init__(s):
        super().__init__()
        s.lin = torch.nn.Linear(5, 1); s.embed = SNW(5000, 6)
class TM2(torch.nn.Module):
    def __init__(s):
        super().__init__()
        s.embedder = torch.nn.Linear(3, 2); s.generator = torch.nn.Linear(4, 3); s.discriminator = Dis()
torch.manual_seed(1 + rank)                                  # different (u, v) per rank: the constructor's broadcast must unify them
tm2 = TM2()
oD2 = Arena(tm2.discriminator.parameters())
red3 = GradReducer(tm2, finetune=False, optimizer_D=oD2)
u, v = tm2.discriminator.embed.weight_u, tm2.discriminator.embed.weight_v
gu = [None] * W; dist.all_gather_object(gu, (u.tolist(), v.tolist()))
ok &= all(g_ == gu[0] for g_ in gu)
g = torch.Generator().manual_seed(100 + rank)
B, E = 3, 6
label = torch.randint(0, 5000, (B,), generator=g); rows = torch.randn(B, E, generator=g); coef = torch.randn((), generator=g)
dense = torch.zeros(5000, E); dense.index_add_(0, label, rows); dense.addmm_((u * (-coef))[:, None], v[None, :])
oD2.flat.fill_(float(rank + 1))
emb_grad = tm2.discriminator.embed.weight_orig.grad
emb_grad.fill_(float('nan'))                              # the sparse path must REBUILD this slice from the published parts (a dense all-reduce would keep NaN)
tm2.discriminator._embed_parts = {'parts': (label, rows, coef, u, v)}
red3.reduce_discriminator_side()
alld = [None] * W; dist.all_gather_object(alld, dense.tolist())
want = sum(torch.tensor(d_) for d_ in alld) / W
ok &= bool(torch.allclose(emb_grad, want, atol=1e-5))
ok &= all(torch.allclose(p.grad, torch.full_like(p, mean)) for p in tm2.discriminator.lin.parameters())
# ragged last batch (ADVICE r03): rank 0 brings 3 rows, every other rank 2 -- the collective sequence and sizes must not depend on the
# rank-local row count (a rank-local guard collective would mismatch and hang); the capacity (3) was agreed on at the first exchange
Br = B if rank == 0 else B - 1
label2 = torch.randint(0, 5000, (Br,), generator=g); rows2 = torch.randn(Br, E, generator=g); coef2 = torch.randn((), generator=g)
dense2 = torch.zeros(5000, E); dense2.index_add_(0, label2, rows2); dense2.addmm_((u * (-coef2))[:, None], v[None, :])
oD2.flat.fill_(float(rank + 1)); emb_grad.fill_(float('nan'))
tm2.discriminator._embed_parts = {'parts': (label2, rows2, coef2, u, v)}
red3.reduce_discriminator_side()
alld = [None] * W; dist.all_gather_object(alld, dense2.tolist())
want = sum(torch.tensor(d_) for d_ in alld) / W
ok &= bool(torch.allclose(emb_grad, want, atol=1e-5)) and red3.max_batch == B
ok &= all(torch.allclose(p.grad, torch.full_like(p, mean)) for p in tm2.discriminator.lin.parameters())
# (ADVICE r04) a rank WITHOUT sparse parts at the first exchange must not skip a collective the others issue: the first call agrees on
# "row-sparse or dense" with one all-reduce on EVERY rank; here only rank 0 has parts, so all ranks fall back to the dense all-reduce
torch.manual_seed(7)
tm4 = TM2(); oD4 = Arena(tm4.discriminator.parameters())
red6 = GradReducer(tm4, finetune=False, optimizer_D=oD4)
oD4.flat.fill_(float(rank + 1))
tm4.discriminator._embed_parts = {'parts': (label, rows, coef, u, v)} if rank == 0 else {}
red6.reduce_discriminator_side()
ok &= red6.use_sparse is False and bool(torch.allclose(oD4.flat, torch.full_like(oD4.flat, mean)))
print('REDUCER_OK' if ok else 'REDUCER_FAIL', flush=True)
dist.destroy_process_group()
'''


@pytest.mark.parametrize('world', [2, 4])
def test_grad_reducer_gloo(tmp_path, world):
    """parallel.GradReducer on 2 and 4 gloo ranks: ONE flat start-up broadcast (parameters + the label embedding's power-iteration
    vectors), per-side mean all-reduce (tensor list and flat-arena paths), the row-sparse label-embedding exchange with int64 labels"""
    script = tmp_path / 'worker.py'
    script.write_text(REDUCER_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29533 + world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all('REDUCER_OK' in o for o in outs), outs


def test_radam_state_loads_into_a_reference_style_radam():
    """the reference's RAdam indexes ``group['buffer'][step % 10]`` in step() (utils/radam.py:63): an optimizer state written here must
    carry that per-group cache, for the restated RAdam and for the fused one (whose state_dict needs no GPU)"""
    import torch
    from latent_pose_reenactment_amd.optim import FusedRAdam
    from latent_pose_reenactment_amd.utils.radam import RAdam
    for cls in (RAdam, FusedRAdam):
        p = torch.nn.Parameter(torch.zeros(3))
        sd = cls([p], lr=5e-4, betas=(0.0, 0.999), eps=1e-5).state_dict()
        buf = sd['param_groups'][0]['buffer']
        assert len(buf) == 10 and all(len(b) == 3 for b in buf), cls
    # and a state with the cache (as the reference writes it) loads back
    opt = RAdam([torch.nn.Parameter(torch.zeros(3))])
    opt.load_state_dict(sd)
    assert 'buffer' in opt.param_groups[0]


def test_second_accumulation_buffers_fold_into_grad_once():
    """nn.alt_accumulation (round 4): a backward pass that runs beside another one feeding the same parameters adds into per-parameter SECOND
    buffers; leaving ``fused_grad_accumulation`` folds them into .grad exactly once and leaves them zero for the next step"""
    import torch
    from latent_pose_reenactment_amd import nn as lpnn
    w = torch.nn.Parameter(torch.zeros(3, 4))
    w.grad = torch.full((3, 4), 1.0)
    assert lpnn._accum_target(w) is None                      # outside the context: plain autograd accumulation
    with lpnn.fused_grad_accumulation():
        direct = lpnn._accum_target(w)
        assert direct is w.grad
        alt = lpnn._accum_target(w, alt=True)
        assert alt is not w.grad and float(alt.abs().sum()) == 0.0
        direct += 2.0                                        # "real-image pass": straight into .grad
        alt += 5.0                                           # "fake-image pass" on the other stream: second buffer
        assert torch.equal(w.grad, torch.full((3, 4), 3.0))
    assert torch.equal(w.grad, torch.full((3, 4), 8.0))     # folded in on exit
    assert float(alt.abs().sum()) == 0.0 and not lpnn._ALT['dirty']
    with lpnn.fused_grad_accumulation():
        assert lpnn._accum_target(w, alt=True) is alt          # persistent buffer (static address: hipGraph friendly)
    assert torch.equal(w.grad, torch.full((3, 4), 8.0))     # nothing added twice
    #  a backward pass that raises inside the context must not leak its partial second-buffer gradients into the next step
    with pytest.raises(ZeroDivisionError):
        with lpnn.fused_grad_accumulation():
            lpnn._accum_target(w, alt=True).add_(7.0)
            1 / 0
    assert float(alt.abs().sum()) == 0.0 and not lpnn._ALT['dirty'] and torch.equal(w.grad, torch.full((3, 4), 8.0))
    with lpnn.fused_grad_accumulation():
        pass
    assert torch.equal(w.grad, torch.full((3, 4), 8.0))
    # forward-time marker
    assert lpnn._ALT['on'] is False
    with lpnn.alt_accumulation():
        assert lpnn._ALT['on'] is True
    assert lpnn._ALT['on'] is False


def test_resnext_operand_modes_per_block_and_per_contraction(monkeypatch):
    """backbones.ResNeXt.block_precs / layer_precs: default assignment under the global fp16 mode = bf16x3 head + fp16 tail of 6 blocks; the
    LP_E_HEAD_F16 experiment knob moves single KINDS of contractions of the head to fp16 (empty by default)"""
    from embedders import backbones
    from latent_pose_reenactment_amd.nn import PREC_NAMES
    for k in ('LP_PREC', 'LP_PREC_E', 'LP_E_F16_TAIL', 'LP_E_HEAD_F16'):
        monkeypatch.delenv(k, raising=False)
    net = backbones.resnext50_32x4d(num_classes=8)
    x3, f16 = PREC_NAMES['bf16x3'], PREC_NAMES['f16']
    if net.prec != x3:
        import pytest
        pytest.skip('the process was started with another encoder mode')
    bp = net.block_precs()
    assert len(bp) == 16 and bp == [x3] * 10 + [f16] * 6
    lp = net.layer_precs()
    assert all(lp[(b[0], k)] == p for b, p in zip(net._hip_blocks, bp) for k in ('conv1', 'conv2', 'conv3'))
    monkeypatch.setenv('LP_E_HEAD_F16', 'conv2')
    lp = net.layer_precs()
    first, last = net._hip_blocks[0][0], net._hip_blocks[-1][0]
    assert lp[(first, 'conv1')] == x3 and lp[(first, 'conv2')] == f16 and lp[(first, 'conv3')] == x3
    assert lp[(last, 'conv1')] == lp[(last, 'conv2')] == f16
    monkeypatch.setenv('LP_E_F16_TAIL', '0')
    assert net.block_precs() == [x3] * 16


def test_roctx_ranges_are_a_noop_when_off_and_bind_libroctx_when_on():
    """utils/tracing.py (SURVEY 5 tracing row): LP_ROCTX=1 brackets the step's phases with roctxRangePushA / roctxRangePop of the ROCm installation
    (ctypes; no GPU needed to push a range), and without the variable ``rng`` costs one generator frame"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\nfrom latent_pose_reenactment_amd.utils import tracing\n"
            "with tracing.rng('probe'):\n    pass\nprint('enabled', tracing.enabled())") % ROOT
    for flag, want in (('0', 'enabled False'), ('1', 'enabled True')):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, LP_ROCTX=flag), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and want in r.stdout, (flag, r.stdout, r.stderr[-500:])


def test_step_breakdown_reads_the_marker_period_off_the_trace(tmp_path):
    """scripts/step_breakdown.py segments a rocprofv3 kernel trace into steps by the optimizer-step markers; a step holds 2 markers (G, D) or 3 (the
    generator's slice of optimizer_G stepped on its own since round 6), the kernels of concurrent streams interleave a little differently from replay to
    replay, and bench.py ends with two eager instrumented steps whose kernel counts differ: the script must still cut out ONE full graph replay"""
    import csv
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for marks_per_step, segs in ((3, (50, 30, 20)), (2, (70, 30))):
        rows, t = [], 0

        def k(name, d=10):
            nonlocal t
            rows.append((t, t + d, name)); t += d + 2
        for step in range(16):
            eager_tail = step >= 14          # the last two steps: other kernel counts (the instrumented eager steps)
            for si, n in enumerate(segs):
                for i in range(n + (step % 2 if si == 0 and not eager_tail else 0) + (7 if eager_tail else 0)):      # +-1 kernel of interleaving jitter
                    k(f'conv_{si}(float*)')
                k('mt_step_inc_kernel(int*)')
        path = tmp_path / f'trace{marks_per_step}.csv'
        with open(path, 'w') as f:
            w = csv.writer(f)
            w.writerow(['Start_Timestamp', 'End_Timestamp', 'Kernel_Name'])
            w.writerows(rows)
        r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'step_breakdown.py'), str(path)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        head = r.stdout.splitlines()[0]
        kernels = int(head.rsplit('kernels', 1)[1])
        assert f'period {marks_per_step}' in r.stderr, r.stderr
        assert abs(kernels - (sum(segs) + marks_per_step - 1)) <= 2, (head, r.stderr)
