#!/usr/bin/env python3
"""bench.py -- train-step images/s at 256x256, per-GPU bs=8 (BASELINE.json metric), on N MI355X GPUs.

ONE workload for every N (so that 1/2/4/8-GPU values form a scaling curve): the META-TRAINING step of configs/default.yaml (BASELINE
configs[2] -- the configuration the reference runs data parallel): ResNeXt-50 identity encoder over 8 frames per sample + MobileNetV2 pose
encoder (both trained), G, D with the 98000 x 512 label embedding, criterions idt_embed, perceptual, adversarial, featmat, dis_embed,
dice, Adam, EMA; bs 8 per GPU, 256 x 256, synthetic VoxCeleb2-shaped batch, random-init weights (no network for datasets/checkpoints).
One "step" = runners/holycow.py:230-257: E -> G -> D x3 -> criterions -> G backward/[all-reduce]/step -> D backward/[all-reduce]/step -> EMA,
every layer of it on the hand-written gfx950 kernels of liblp_hip.so.  N > 1: each rank on its own batch, RCCL gradient all-reduce of
latent_pose_reenactment_amd.parallel (weak scaling).
`--workload finetune_step` = BASELINE configs[1] (finetuning-base.yaml; the reference refuses multi-GPU fine-tuning): at N = 1 the
default run also measures it in a child process and reports it under "finetune_step".

Prints ONE JSON line on rank 0 (see README/DESIGN for the field definitions)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))    # plugin packages: generators/, criterions/, ...

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

GEN_FWD_GFLOP_PER_IMAGE = 60.72      # SURVEY 8d / BASELINE.md: dense 2*MAC, as executed by the reference
MFMA_BF16_PEAK_TFLOPS = 2500.0       # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
# dense 2*MAC per image of the other nets (ResNeXt-50 / MobileNetV2 per FRAME, from the layer shapes at 256 x 256)
DIS_GFLOP, VGG19_GFLOP, VGGFACE_GFLOP, RESNEXT_GFLOP_PER_FRAME, MOBILENET_GFLOP_PER_FRAME = 30.97, 47.34, 40.09, 11.0, 0.8


ROUND = 'r06'          # prefix of the profiles/ files this tree's tests and artifact script write and this file quotes


def source_stamp():
    """what ties a measured file under profiles/ to the tree that produced it: sha256 over the kernel sources + C header (`csrc`) and over
    the package's Python files (`py`), 16 hex digits each -- computable on the GPU box (which has no .git) and here; `head` = git HEAD when
    a repository is present (build container), else None.  bench.py recomputes the stamp at run time and marks a profile file whose stamp
    differs as STALE instead of printing its figures as fact."""
    import hashlib
    import subprocess

    def digest(paths):
        h = hashlib.sha256()
        for p_ in sorted(paths):
            h.update(os.path.relpath(p_, ROOT).encode())
            h.update(hashlib.sha256(open(p_, 'rb').read()).digest())
        return h.hexdigest()[:16]
    pkg = os.path.join(ROOT, 'latent_pose_reenactment_amd')
    csrc = [os.path.join(pkg, 'csrc', f) for f in os.listdir(os.path.join(pkg, 'csrc')) if f.endswith(('.hip', '.h'))]
    csrc.append(os.path.join(ROOT, 'include', 'lp_hip.h'))
    py = []
    for d, dirs, files in os.walk(pkg):
        dirs[:] = [x for x in dirs if x not in ('__pycache__', 'build', '.pytest_cache')]
        py += [os.path.join(d, f) for f in files if f.endswith('.py')]
    head = None
    try:
        r = subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short=12', 'HEAD'], capture_output=True, text=True, timeout=10)
        head = r.stdout.strip() or None if r.returncode == 0 else None
    except Exception:
        pass
    return {'csrc': digest(csrc), 'py': digest(py), 'head': head}


def stamp_status(stamp):
    """-> (stale: bool, why) of a profile file's stamp against the running tree"""
    if not isinstance(stamp, dict):
        return True, 'no source stamp in the file'
    now = source_stamp()
    diff = [k for k in ('csrc', 'py') if stamp.get(k) != now[k]]
    return bool(diff), ('' if not diff else 'source changed since the file was measured: ' + ', '.join(diff))


def step_algorithmic_tflop(workload, batch, frames=8, image_size=256, generator='vector_pose_unsupervised_segmentation_noBottleneck'):
    """useful dense FLOPs of one training step as THIS path executes it (fwd = 1, data gradient = 1, weight gradient = 1 per layer):
    G fwd + both gradients; D: the fake pass of the G loss (fwd + dgrad, its weight gradients are never consumed), the detached fake and
    the real pass of the D loss (fwd + both gradients each); VGG19 / VGGFace: fake fwd + dgrad, real fwd; meta-training adds the encoders
    (fwd + both gradients).  The reference's step also computes D weight gradients in the G backward and discards them (holycow.py:247)."""
    per_img = 3 * GEN_FWD_GFLOP_PER_IMAGE
    if workload == 'generator':
        if generator == 'FSTH_plus' and image_size == 512:
            per_img = 3 * 242.5          # FSTH_plus forward at 512 x 512 (one more up block: not the 256 x 256 count x 4)
        else:
            assert image_size == 256, 'FLOP table holds the 256 x 256 generator and the 512 x 512 FSTH_plus generator (BASELINE.md section 2)'
        return per_img * batch / 1e3
    per_img += (2 + 3 + 3) * DIS_GFLOP + 3 * VGG19_GFLOP + 3 * VGGFACE_GFLOP
    if workload == 'metatrain_step':
        per_img += 3 * (RESNEXT_GFLOP_PER_FRAME * frames + MOBILENET_GFLOP_PER_FRAME)
    return per_img * batch / 1e3


def stream_config(workload):
    """which independent branches of the step run on their own HIP streams (latent_pose_reenactment_amd/streams.py)"""
    from latent_pose_reenactment_amd import streams
    probe = torch.empty(1, device='cuda') if torch.cuda.is_available() else None
    ft = workload != 'metatrain_step'
    on = [k for k in ('encoders', 'criterions', 'prepare', 'dpasses', 'real', 'optimizer') if probe is not None and streams.enabled(probe, k, finetuning=ft) and not (ft and k == 'encoders')]
    on += [k for k in ('ebwd', 'gwgrad') if probe is not None and not ft and streams.enabled(probe, k, finetuning=False) and int(os.environ.get('WORLD_SIZE', '1')) == 1]
    return {'concurrent_branches': on if workload != 'generator' else [],
            'note': 'encoders: pose encoder beside the identity encoder; criterions: VGG-19 / VGGFace stacks beside the discriminator pass; '
                    'prepare: spectral-norm power iterations + weight packs of G and D beside the encoders; dpasses: the discriminator\'s three '
                    'passes beside each other; real: its real-image pass already beside the generator forward; ebwd (one GPU): the encoders\' backward beside loss_D.backward; '
                    'gwgrad (with ebwd): the generator\'s weight-gradient launches issued on the critic-backward stream; autograd runs each backward on its forward stream; captured as parallel '
                    'paths of the hipGraphs'}


def make_args(image_size, batch_size, device, num_gpus, rank, prec_name, finetune=True):
    """finetune=True: configs/finetuning-base.yaml (BASELINE configs[1]); False: configs/default.yaml meta-training (configs[2])"""
    a = argparse.Namespace(
        image_size=image_size, batch_size=batch_size * num_gpus, num_gpus=num_gpus, world_size=num_gpus, rank=rank, device=device,
        in_channels=3, out_channels=3, num_channels=64, max_num_channels=512, embed_channels=512, pose_embedding_size=256,
        gen_padding='zero', norm_layer='in', gen_constant_input_size=4, gen_num_residual_blocks=2,
        dis_padding='zero', dis_num_blocks=7, num_labels=98000, average_function='sum',
        gan_type='gan', fm_weight=10.0, dice_weight=1.0, perc_weight=3e-2, idt_embed_weight=0.6e-2, dis_embed_weight=1e-2,
        vgg_weights_dir='/nonexistent', synthetic_vgg_seed=1234, n_frames_for_encoder=8,
        optimizer='RAdam' if finetune else 'Adam', lr_gen=5e-4 if finetune else 5e-5, lr_dis=8e-4 if finetune else 2e-4, beta1=0.0,
        finetune=finetune, random_seed=123)
    os.environ['LP_PREC'] = prec_name
    return a


def build(args):
    import importlib
    GW = importlib.import_module('generators.' + getattr(args, 'generator', 'vector_pose_unsupervised_segmentation_noBottleneck')).Wrapper
    from embedders.unsupervised_pose_separate_embResNeXt_segmentation import Wrapper as EW
    from discriminators.no_landmarks import Wrapper as DW
    from criterions import adversarial, featmat, idt_embed, perceptual, dice, dis_embed
    from runners import holycow
    torch.manual_seed(args.random_seed)
    D = DW.get_net(args)
    G = GW.get_net(args)
    E = EW.get_net(args)
    if args.finetune:
        crit_modules = (adversarial, featmat, idt_embed, perceptual, dice)                  # configs/finetuning-base.yaml:7
    else:
        crit_modules = (idt_embed, perceptual, adversarial, featmat, dis_embed, dice)       # configs/default.yaml:5
    crits = [m.Wrapper.get_net(args) for m in crit_modules]
    tm = holycow.TrainingModule(E, G, D, crits, [], {})
    if args.finetune:
        # fine-tuning bootstrap (train.py:218-279): identity embedding e_hat -> generator parameter / discriminator row
        e_hat = torch.randn(1, args.embed_channels, device=args.device) * 0.1
        dd = {'embeds': e_hat}
        tm.embedder.enable_finetuning()
        tm.generator.enable_finetuning(dd)
        tm.discriminator.enable_finetuning(dd)
        tm.running_averages['generator'].enable_finetuning(dd)
        tm.running_averages['embedder'].enable_finetuning()
    opt_G = holycow.get_optimizer(tm.embedder, tm.generator, args)
    opt_D = DW.get_optimizer(tm.discriminator, args)
    tm.train()
    return tm, opt_G, opt_D, holycow


def synthetic_batch(args, per_gpu_batch, seed):
    from dataloaders.synthetic_voxceleb2 import make_sample
    frames = 1 if args.finetune else args.n_frames_for_encoder
    datas, targets = zip(*[make_sample(i, args.image_size, frames, args.num_labels, args.finetune, seed) for i in range(per_gpu_batch)])
    data = {k: torch.stack([d[k] for d in datas]).to(args.device) for k in datas[0]}
    target = {'real_segm': torch.stack([t['real_segm'] for t in targets]).to(args.device),
              'label': torch.tensor([t['label'] for t in targets], device=args.device)}
    return data, target


def _cpu_step_metatrain(args, sample_batch):
    """-> a callable running ONE meta-training step (configs/default.yaml) of `sample_batch` samples on the CPU: ResNeXt-50 over the
    8 encoder frames + MobileNetV2 (the torchvision-compatible containers of embedders/backbones.py evaluated by oracle/backbones_ref.py:
    stock torch-CPU layers, train mode) -> oracle generator -> oracle discriminator x3 with the 98000 x 512 label embedding -> VGGFace / VGG19 / adversarial /
    featmat / dis_embed / dice -> loss_G.backward -> Adam(G + E) -> loss_D.backward -> Adam(D) -> EMA(G + E)."""
    import copy
    from oracle import lp_oracle as O
    from oracle import backbones_ref as BR
    from generators.vector_pose_unsupervised_segmentation_noBottleneck import Wrapper as GW
    from discriminators.no_landmarks import Wrapper as DW
    from embedders.backbones import mobilenet_v2, resnext50_32x4d
    from criterions.common.perceptual_loss import PerceptualLoss
    from dataloaders.synthetic_voxceleb2 import make_sample
    a = copy.copy(args)
    a.device = 'cpu'
    torch.manual_seed(0)
    G, D = GW.get_net(a), DW.get_net(a)
    idt_net, pose_net = resnext50_32x4d(a.embed_channels).train(), mobilenet_v2(a.pose_embedding_size).train()
    vgg19 = PerceptualLoss(a.perc_weight, '/nonexistent', 'caffe', synthetic_seed=1234).model.state_dict()
    vggf = PerceptualLoss(a.idt_embed_weight, '/nonexistent', 'face', synthetic_seed=1235).model.state_dict()

    def as_sd(module):
        sd = {k: v.detach().clone() for k, v in module.state_dict().items()}
        params = [k for k, _ in module.named_parameters()]
        for k in params:
            sd[k].requires_grad_(True)
        return sd, params
    sdG, pG = as_sd(G)
    sdD, pD = as_sd(D)
    e_params = list(idt_net.parameters()) + list(pose_net.parameters())
    ema = {k: sdG[k].detach().clone() for k in pG}
    ema_e = [p.detach().clone() for p in e_params]
    mom = {id(sd): {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in ps} for sd, ps in ((sdG, pG), (sdD, pD))}
    mom_e = [(torch.zeros_like(p), torch.zeros_like(p)) for p in e_params]
    data, target = zip(*[make_sample(i, a.image_size, a.n_frames_for_encoder, a.num_labels, False, 7) for i in range(sample_batch)])
    enc = torch.stack([d['enc_rgbs'] for d in data])                      # B x K x 3 x S x S
    pose_in = torch.stack([d['pose_input_rgbs'][0] for d in data])
    tgt = torch.stack([d['target_rgbs'][0] for d in data])
    real_segm = torch.stack([t['real_segm'] for t in target])
    label = torch.tensor([t['label'] for t in target], dtype=torch.long)
    step_no = [0]

    def one():
        step_no[0] += 1
        b, k = enc.shape[:2]
        per_frame = BR.resnext_forward(idt_net, enc.reshape(b * k, *enc.shape[2:])).view(b, k, -1)
        embeds = per_frame.mean(1)
        pose = BR.mobilenet_forward(pose_net, pose_in)
        rgb, segm = O.generator_forward(sdG, embeds, pose, num_channels=64, max_num_channels=512, image_size=a.image_size, train=True)
        out = O.discriminator_forward(sdD, rgb, tgt, label, image_size=a.image_size, dis_num_blocks=7, train=True, embed_eps=O.SN_EPS_CONV)
        lg, ld = O.adversarial_gan(out['fake_score_G'], out['fake_score_D'], out['real_score'])
        loss_G = lg + O.feature_matching(out['fake_features'], out['real_features']) \
            + O.perceptual_loss(vggf, O.crop_and_resize_fixed(rgb), O.crop_and_resize_fixed(tgt), a.idt_embed_weight, O.VGG16_CFG) \
            + O.perceptual_loss(vgg19, rgb, tgt, a.perc_weight, O.VGG19_CFG) + O.dice(segm, real_segm) \
            + O.dis_embed(per_frame, out['real_embedding'], a.dis_embed_weight)
        gs = torch.autograd.grad(loss_G, [sdG[k] for k in pG] + e_params, retain_graph=True, allow_unused=True)
        with torch.no_grad():
            for k, g in zip(pG, gs):
                if g is not None:
                    O.adam_step(sdG[k], g, *mom[id(sdG)][k], step_no[0], a.lr_gen, 0.0, 0.999, 1e-5)
            for p_, g, m_ in zip(e_params, gs[len(pG):], mom_e):
                if g is not None:
                    O.adam_step(p_, g, *m_, step_no[0], a.lr_gen, 0.0, 0.999, 1e-5)
        gD = torch.autograd.grad(ld, [sdD[k] for k in pD], allow_unused=True)
        with torch.no_grad():
            for k, g in zip(pD, gD):
                if g is not None:
                    O.adam_step(sdD[k], g, *mom[id(sdD)][k], step_no[0], a.lr_dis, 0.0, 0.999, 1e-5)
            for k in pG:
                O.ema_update(ema[k], sdG[k], 0.999)
            for av, p_ in zip(ema_e, e_params):
                O.ema_update(av, p_.detach(), 0.999)
    return one


def _cpu_step(args, sample_batch):
    """-> a callable running ONE fine-tuning step of `sample_batch` images through the oracle (oracle/lp_oracle.py):
    pose encoder -> generator -> discriminator x3 -> adversarial/featmat/VGGFace/VGG19/dice -> loss_G.backward -> RAdam(G) ->
    loss_D.backward -> RAdam(D) -> EMA(G)."""
    import copy
    from oracle import lp_oracle as O
    from oracle import backbones_ref as BR
    from generators.vector_pose_unsupervised_segmentation_noBottleneck import Wrapper as GW
    from discriminators.no_landmarks import Wrapper as DW
    from embedders.backbones import mobilenet_v2
    from criterions.common.perceptual_loss import PerceptualLoss
    from dataloaders.synthetic_voxceleb2 import make_sample
    a = copy.copy(args)
    a.device = 'cpu'
    torch.manual_seed(0)
    G, D, pose_net = GW.get_net(a), DW.get_net(a), mobilenet_v2(256)
    e_hat = torch.randn(1, 512) * 0.1
    G.enable_finetuning({'embeds': e_hat}); D.enable_finetuning({'embeds': e_hat})
    vgg19 = PerceptualLoss(a.perc_weight, '/nonexistent', 'caffe', synthetic_seed=1234).model.state_dict()
    vggf = PerceptualLoss(a.idt_embed_weight, '/nonexistent', 'face', synthetic_seed=1235).model.state_dict()

    def as_sd(module):
        sd = {k: v.detach().clone() for k, v in module.state_dict().items()}
        params = [k for k, _ in module.named_parameters()]
        for k in params:
            sd[k].requires_grad_(True)
        return sd, params
    sdG, pG = as_sd(G)
    sdD, pD = as_sd(D)
    ema = {k: sdG[k].detach().clone() for k in pG}
    mom = {id(sd): {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in ps} for sd, ps in ((sdG, pG), (sdD, pD))}
    data, target = zip(*[make_sample(i, a.image_size, 1, a.num_labels, True, 7) for i in range(sample_batch)])
    pose_in = torch.stack([d['pose_input_rgbs'][0] for d in data])
    tgt = torch.stack([d['target_rgbs'][0] for d in data])
    real_segm = torch.stack([t['real_segm'] for t in target])
    label = torch.zeros(sample_batch, dtype=torch.long)
    step_no = [0]

    def one():
        step_no[0] += 1
        with torch.no_grad():
            pose = BR.mobilenet_forward(pose_net, pose_in)
        rgb, segm = O.generator_forward(sdG, sdG['identity_embedding'], pose, num_channels=64, max_num_channels=512,
                                        image_size=a.image_size, train=True)
        out = O.discriminator_forward(sdD, rgb, tgt, label, image_size=a.image_size, dis_num_blocks=7, train=True,
                                      embed_eps=O.SN_EPS_DEFAULT)
        lg, ld = O.adversarial_gan(out['fake_score_G'], out['fake_score_D'], out['real_score'])
        loss_G = lg + O.feature_matching(out['fake_features'], out['real_features']) \
            + O.perceptual_loss(vggf, O.crop_and_resize_fixed(rgb), O.crop_and_resize_fixed(tgt), a.idt_embed_weight, O.VGG16_CFG) \
            + O.perceptual_loss(vgg19, rgb, tgt, a.perc_weight, O.VGG19_CFG) + O.dice(segm, real_segm)
        gG = torch.autograd.grad(loss_G, [sdG[k] for k in pG], retain_graph=True, allow_unused=True)
        with torch.no_grad():
            for k, g in zip(pG, gG):
                if g is not None:
                    O.radam_step(sdG[k], g, *mom[id(sdG)][k], step_no[0], a.lr_gen, 0.0, 0.999, 1e-5)
        gD = torch.autograd.grad(ld, [sdD[k] for k in pD], allow_unused=True)
        with torch.no_grad():
            for k, g in zip(pD, gD):
                if g is not None:
                    O.radam_step(sdD[k], g, *mom[id(sdD)][k], step_no[0], a.lr_dis, 0.0, 0.999, 1e-5)
            for k in pG:
                O.ema_update(ema[k], sdG[k], 0.972)
    return one


def _cpu_info():
    model, phys = 'unknown', set()
    try:
        cur = {}
        for line in open('/proc/cpuinfo'):
            if ':' in line:
                k, v = [t.strip() for t in line.split(':', 1)]
                if k == 'model name':
                    model = v
                cur[k] = v
            elif not line.strip() and cur:
                phys.add((cur.get('physical id', '0'), cur.get('core id', cur.get('processor', '0'))))
                cur = {}
    except OSError:
        pass
    return model, len(phys) or (os.cpu_count() or 1), os.cpu_count() or 1


def _median_time(fn, warm, reps, budget_s):
    """median of up to `reps` timed calls after `warm` warm-up calls, inside a wall-clock budget: when the warm-up alone exhausts the
    budget its last call is the (single) sample"""
    t_all = time.time()
    last = None
    for _ in range(warm):
        t0 = time.time(); fn(); last = time.time() - t0
    ts = []
    while len(ts) < reps and (time.time() - t_all < budget_s or not ts):      # always at least ONE timed (warm) step
        t0 = time.time(); fn(); ts.append(time.time() - t0)
    if not ts:
        return last, 1
    ts.sort()
    return ts[len(ts) // 2], len(ts)


def _numa_nodes():
    """[(node, [ONE logical cpu per physical core of that node])] from sysfs -- the placement `numactl --cpunodebind / --membind` would give
    (numactl itself is not in the image; first-touch allocation by the pinned threads keeps the memory node-local)"""
    import glob
    import re

    def cpus(txt):
        out = []
        for part in txt.strip().split(','):
            if '-' in part:
                a, b = part.split('-'); out += list(range(int(a), int(b) + 1))
            elif part:
                out.append(int(part))
        return out
    nodes = []
    allowed = set(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None
    for d in sorted(glob.glob('/sys/devices/system/node/node[0-9]*'), key=lambda x: int(re.findall(r'(\d+)$', x)[0])):
        try:
            lst = cpus(open(os.path.join(d, 'cpulist')).read())
        except OSError:
            continue
        seen, pick, every = set(), [], []
        for c in lst:
            if allowed is not None and c not in allowed:
                continue
            every.append(c)
            try:
                sib = tuple(cpus(open(f'/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list').read()))
            except OSError:
                sib = (c,)
            if sib not in seen:
                seen.add(sib); pick.append(c)
        if pick:
            nodes.append((int(re.findall(r'(\d+)$', d)[0]), pick, every))
    return nodes


def _cpu_generator_only(args, sample_batch):
    """-> a callable: generator forward + backward of `sample_batch` images through the oracle (the generator-only row;
    the same white-noise loss as `bench.py --workload generator`)"""
    import copy
    from oracle import lp_oracle as O
    from generators.vector_pose_unsupervised_segmentation_noBottleneck import Wrapper as GW
    a = copy.copy(args)
    a.device = 'cpu'
    torch.manual_seed(0)
    G = GW.get_net(a)
    sd = {k: v.detach().clone() for k, v in G.state_dict().items()}
    params = [k for k, _ in G.named_parameters()]
    for k in params:
        sd[k].requires_grad_(True)
    e, p = torch.randn(sample_batch, a.embed_channels), torch.randn(sample_batch, a.pose_embedding_size)

    def one():
        rgb, segm = O.generator_forward(sd, e, p, num_channels=64, max_num_channels=512, image_size=a.image_size, train=True)
        torch.autograd.grad(rgb.mean() + segm.mean(), [sd[k] for k in params], allow_unused=True)
    return one


def _cpu_drive_frame(args):
    """-> a callable: ONE frame of the drive.py loop (drive.py:84-88): MobileNetV2 pose encoder + generator, B = 1, eval mode, no autograd"""
    import copy
    from oracle import lp_oracle as O
    from oracle import backbones_ref as BR
    from generators.vector_pose_unsupervised_segmentation_noBottleneck import Wrapper as GW
    from embedders.backbones import mobilenet_v2
    a = copy.copy(args)
    a.device = 'cpu'
    torch.manual_seed(0)
    G, pose_net = GW.get_net(a), mobilenet_v2(a.pose_embedding_size).eval()
    G.enable_finetuning({'embeds': torch.randn(1, a.embed_channels) * 0.1})
    sd = {k: v.detach().clone() for k, v in G.state_dict().items()}
    frame = torch.rand(1, 3, a.image_size, a.image_size)

    def one():
        with torch.no_grad():
            pose = BR.mobilenet_forward(pose_net, frame)
            O.generator_forward(sd, sd['identity_embedding'], pose, num_channels=64, max_num_channels=512, image_size=a.image_size, train=False)
    return one


def cpu_worker(spec):
    """child process of cpu_baseline: `threads,batch,warm,reps,budget,image_size,kind` -> one JSON line {"t": median seconds per step, "n": timed steps}
    (its own process so that OMP_NUM_THREADS / torch.set_num_threads / the CPU affinity take effect before any CPU kernel has run).
    kind 0: fine-tuning step, 1: meta-training step, 2: generator forward + backward only, 3: one drive.py frame.
    LP_CPU_AFFINITY = comma-separated logical cpus: pin this worker (one worker per NUMA node)."""
    threads, batch, warm, reps, budget, image_size, kind = [int(float(v)) for v in spec.split(',')]
    aff = os.environ.get('LP_CPU_AFFINITY')
    if aff and hasattr(os, 'sched_setaffinity'):
        os.sched_setaffinity(0, {int(c) for c in aff.split(',')})
    torch.set_num_threads(threads)
    args = make_args(image_size, 8, 'cpu', 1, 0, 'bf16x3', finetune=kind != 1)
    fn = {0: lambda: _cpu_step(args, batch), 1: lambda: _cpu_step_metatrain(args, batch), 2: lambda: _cpu_generator_only(args, batch),
          3: lambda: _cpu_drive_frame(args)}[kind]()
    t, n = _median_time(fn, warm, reps, budget)
    print(json.dumps({'t': t, 'n': n, 'threads': torch.get_num_threads()}))


def cpu_baseline(args, full=False, workload='metatrain_step'):
    """The CPU path timed on this box's host cores: the parity-pinned fp32 torch-CPU restatement of the reference
    (oracle/lp_oracle.py; for the encoders the torchvision-compatible stock layers) running the SAME training step on the SAME kind of
    synthetic batch, each row in its own process(es).
      all-core row : EVERY physical core, the bs = 8 batch of the GPU step: one worker process per NUMA node, pinned to that node's physical
                     cores (sched_setaffinity: what numactl --cpunodebind / --membind does), each stepping its share of the 8 samples
                     concurrently; value = 8 samples / the slowest worker's median step time.  (Round 3 ran ONE process over 64 of 128
                     cores on 2 samples: torch-CPU scales badly across sockets from a single process.)
      1-thread row : how the reference configures itself (OMP_NUM_THREADS=1, torch.set_num_threads(1): train.py:2, utils/utils.py:19), 1 sample.
    Default = a bounded sample (1 warm-up + up to 3 timed all-core steps, 1 + up to 2 one-thread steps: 1-2 minutes);
    --cpu-baseline-full runs the >= 3 warm-up + >= 10 timed protocol."""
    import subprocess
    model, cores, logical = _cpu_info()
    warm, reps = (3, 10) if full else (1, 3)
    meta = int(workload == 'metatrain_step')
    name = 'meta-training' if meta else 'fine-tuning'
    nodes = _numa_nodes() or [(0, list(range(cores)), list(range(logical)))]
    while len(nodes) > 8 or (8 % len(nodes)):          # the 8 samples must split evenly: merge neighbouring nodes
        nodes = [(nodes[i][0], nodes[i][1] + (nodes[i + 1][1] if i + 1 < len(nodes) else []), nodes[i][2] + (nodes[i + 1][2] if i + 1 < len(nodes) else []))
                 for i in range(0, len(nodes), 2)]
    per = 8 // len(nodes)

    def spawn(threads, batch, w, r, budget, cpus=None, kind=None):
        # OMP_WAIT_POLICY=passive: the autograd engine's thread runs its own OpenMP team beside the forward thread's; with active waiting
        # the two teams of a pinned worker spin against each other on the same cores (measured: 145 - 207 s per step instead of ~20 s)
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), OMP_WAIT_POLICY='passive', GOMP_SPINCOUNT='0')
        if cpus:
            env['LP_CPU_AFFINITY'] = ','.join(str(c) for c in cpus)
        #  stderr goes to /dev/null: the workers are waited for one at a time, and a later one that filled a 64 KB stderr pipe
        # with warnings would block -- and be timed -- until its turn
        return subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', f'{threads},{batch},{w},{r},{budget},{args.image_size},{kind if kind is not None else meta}'],
                                env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)

    def result(proc, budget):
        out, _ = proc.communicate(timeout=budget * 4 + 900)
        return json.loads([l for l in out.splitlines() if l.startswith('{')][-1])
    budget = 600 if full else 60
    procs = [spawn(len(phys), per, warm, reps, budget, every) for _, phys, every in nodes]      # threads = physical cores; mask = their SMT siblings too
    rs = [result(p_, budget) for p_ in procs]
    t_all = max(r['t'] for r in rs)
    used = sum(len(phys) for _, phys, _e in nodes)
    o = result(spawn(1, 1, warm if full else 1, reps if full else 2, 900 if full else 40), 900 if full else 40)
    # BASELINE.md section 3 also asks for the generator-only fwd+bwd and the drive.py single frame on the host cores: one process on the cores of
    # ONE NUMA node (a B = 8 generator batch does not split over processes without changing nothing but the bookkeeping; a frame is B = 1),
    # and the frame also with 1 thread, the reference's own setting
    extra = {}
    try:
        node0 = nodes[0]
        # (--cpu-baseline-full: the section-3 protocol for these two rows as well, 3 warm-up + 10 timed)
        g_ = result(spawn(len(node0[1]), 8, 3 if full else 1, 10 if full else 2, 240 if full else 30, node0[2], kind=2), 240 if full else 30)
        d_ = result(spawn(len(node0[1]), 1, 3 if full else 1, 10 if full else 3, 60 if full else 20, node0[2], kind=3), 60 if full else 20)
        d1 = result(spawn(1, 1, 3 if full else 1, 10 if full else 2, 90 if full else 20, kind=3), 90 if full else 20)
        extra = {'generator_only': {'value': round(8 / g_['t'], 3), 'unit': 'images/s', 'cores': len(node0[1]), 's_per_step': round(g_['t'], 3),
                                    'sample': f"generator forward + backward, bs 8 at {args.image_size}x{args.image_size}, oracle/lp_oracle.py on the {len(node0[1])} physical cores "
                                              f"of NUMA node {node0[0]}; median of {g_['n']} step(s)"},
                 'drive_frame': {'value': round(1 / d_['t'], 3), 'unit': 'frames/s', 'cores': len(node0[1]), 'ms_per_frame': round(d_['t'] * 1e3, 1),
                                 'one_thread': {'value': round(1 / d1['t'], 3), 'unit': 'frames/s', 'cores': 1, 'ms_per_frame': round(d1['t'] * 1e3, 1)},
                                 'sample': f"drive.py:84-88 loop body (MobileNetV2 stock layers + oracle generator, eval, B = 1, {args.image_size}x{args.image_size}); "
                                           f"median of {d_['n']} / {d1['n']} frame(s)"}}
    except Exception as ex:
        extra = {'generator_only': {'error': repr(ex)}}
    return {'value': round(8 / t_all, 4), 'unit': 'images/s', 'cores': used, 'kind': 'port', 'cpu_model': model, 'physical_cores': cores,
            'value_is': 'aggregate throughput of the concurrent per-NUMA-node workers (each steps its share of the 8 samples: per-worker BatchNorm / dice '
                        'statistics, no gradient exchange between them) -- the most favourable reading of "all cores" for the CPU',
            **extra,
            'logical_cpus': logical, 'workload': workload, 'numa_workers': [{'node': n_, 'cores': len(c_), 'samples': per, 's_per_step': round(r['t'], 3), 'timed_steps': r['n']}
                                                                            for (n_, c_, _e), r in zip(nodes, rs)],
            'one_thread': {'value': round(1 / o['t'], 4), 'unit': 'images/s', 'cores': 1,
                           'sample': f"median of {o['n']} {name} step(s) of 1 sample at {args.image_size}x{args.image_size}, OMP_NUM_THREADS=1 / "
                                     f"torch.set_num_threads(1): {o['t']:.2f} s per step"},
            'sample': f"the bs = 8 {name} step at {args.image_size}x{args.image_size} through oracle/lp_oracle.py (+ the stock torch-CPU encoder layers), torch CPU fp32, "
                      f"on all {used} physical cores of the host ({model}): {len(nodes)} worker process(es), one per NUMA node, pinned to the node's physical cores, "
                      f"{per} samples each, concurrently; median of {min(r['n'] for r in rs)} step(s) per worker, slowest worker {t_all:.2f} s per step"
                      + ('' if full else '; bounded sample -- the >= 3 + >= 10 protocol is `bench.py --cpu-baseline-full` (profiles/)')}


def drive_fps(args, frames=60, batch=1):
    """drive.py hot loop (drive.py:84-88): per frame pose encoder + generator, eval mode, replayed as one hipGraph; ``batch`` driving frames
    per iteration (1: the reference's loop; 8: drive.py --batch_size 8 of this package, ``drive_frames``)."""
    from generators.vector_pose_unsupervised_segmentation_noBottleneck import Wrapper as GW
    from embedders.unsupervised_pose_separate_embResNeXt_segmentation import Wrapper as EW
    torch.manual_seed(5)
    G, E = GW.get_net(args), EW.get_net(args)
    G.enable_finetuning({'embeds': torch.randn(1, args.embed_channels, device=args.device) * 0.1})
    E.enable_finetuning()
    G.eval(); E.eval()
    frame = torch.rand(batch, 1, 3, args.image_size, args.image_size, device=args.device)
    out = {}

    def one():
        d = {'pose_input_rgbs': frame}
        E.get_pose_embedding(d)
        G(d)
        out['rgb'] = d['fake_rgbs']
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                one()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        step = one
        mode = 'eager'
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                one()
            step, mode = g.replay, 'hipgraph'
        except Exception as ex:
            print(f'[bench] drive graph capture failed ({ex!r}); eager', file=sys.stderr)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {'value': round(frames * batch / dt, 2), 'unit': 'frames/s', 'ms_per_frame': round(dt / (frames * batch) * 1e3, 3), 'ms_per_iteration': round(dt / frames * 1e3, 3), 'batch': batch,
            'launch_mode': mode, 'note': 'drive.py:84-88 loop (MobileNetV2 pose encoder + generator on the HIP kernels, eval mode, '
                                          '16-bit weight packs cached), frames resident in HBM'}


def assignment_description(tm):
    """the DEFAULT precision assignment as the built modules report it (LP_PREC=f16: operand mode per net; accumulation fp32 everywhere)"""
    try:
        import sys as _s
        from latent_pose_reenactment_amd import nn as lpnn
        names = {v: k for k, v in lpnn.PREC_NAMES.items()}
        from discriminators.no_landmarks import dpass_prec, gpass_prec
        gp, dp = gpass_prec(), dpass_prec()
        return (f'generator: {names[tm.generator.prec]}; critic: fake -> G pass {names[gp[0]]} from unit {gp[1]}, D-side passes {names[dp[0]]} from unit {dp[1]} '
                f'(unit 0 = stem, 1.. = blocks; fp16 before); VGG19 / VGGFace stacks: {names[lpnn.default_prec()]}')
    except Exception as ex:
        return f'(assignment unavailable: {ex!r})'


def encoder_modes(tm):
    try:
        m = tm.embedder.identity_encoder.block_precs()
        n16 = sum(1 for v in m if v == 2)
        head = {0: 'bf16', 1: 'bf16x3 (3 MFMAs per MAC)', 2: 'f16'}[m[0]]
        return f'; identity encoder: {head} operands in its first {len(m) - n16} bottlenecks + stem, fp16 in the last {n16} (train-mode BatchNorm conditioning); pose encoder: bf16x3'
    except Exception:
        return ''


def _load_profile(name):
    """-> (dict | None, stale flag, reason) of a measured file under profiles/ (``source_stamp`` of the tree that produced it inside)"""
    path = os.path.join(ROOT, 'profiles', name)
    try:
        res = json.load(open(path))
    except Exception:
        return None, True, f'profiles/{name} not present'
    stale, why = stamp_status(res.get('stamp'))
    return res, stale, why


def measured_parity(mode, workload):
    """the parity statement of a precision assignment: MEASURED figures, never a literal.
      forward  : tests/test_metatrain_full_gpu.py -- the full configs[2] forward (256 x 256, 8 x 8 encoder frames, 98000 labels) against fp64
                 stock encoders + the CPU oracle; every figure is the PLAIN rel-L2 |a - b| / |b| (the conditioned figure of the projection
                 score is an extra column);
      gradients: tests/test_full_size_parity.py (G, D, VGG19, VGGFace at 256 x 256: tie-masked against the oracle) and
                 tests/test_e1_full_gpu.py (identity encoder, 64 frames: all-gradient rel-L2 / cosine vs fp64 with the stock-fp32 calibration).
    The files carry the source stamp of the tree that produced them; a file whose stamp differs from the running tree is reported STALE."""
    res, stale, why = _load_profile(f'{ROUND}_parity_configs2_{mode}.json')
    if res is None:
        return {'status': 'unmeasured', 'note': why + ': run tests/test_metatrain_full_gpu.py with LP_PARITY_OUT=profiles'}
    worst = max(res['errors'].items(), key=lambda kv: kv[1])
    out = {'status': 'stale' if stale else 'measured', 'stale': stale, 'source': f'tests/test_metatrain_full_gpu.py -> profiles/{ROUND}_parity_configs2_{mode}.json',
           'geometry': res.get('geometry'),
           'identity_encoder_fp16_tail_blocks': res['identity_encoder_blocks'].count('f16') if res['identity_encoder_blocks'][0] != 'f16' else 'all',
           'critic_fake_to_G_pass': res.get('critic_fake_to_G_pass'), 'critic_D_side_passes': res.get('critic_D_side_passes'),
           'generator_operands': res.get('generator_operands'),
           'plain_rel_l2_vs_reference_chain': {k: float(f'{v:.3g}') for k, v in res['errors'].items()},
           'worst': [worst[0], float(f'{worst[1]:.3g}')], 'within_1e-3': bool(worst[1] < 1e-3),
           'conditioned_extra_column': {k: float(f'{v:.3g}') for k, v in (res.get('conditioned') or {}).items()},
           'stock_fp32_encoders_vs_fp64': res.get('stock_fp32_encoders_vs_fp64')}
    if stale:
        out['stale_reason'] = why
    g, gstale, gwhy = _load_profile(f"{ROUND}_parity_gradients_{'f16' if mode == 'default' else mode}.json")
    if g is None:
        out['gradients'] = {'status': 'unmeasured', 'note': gwhy}
    else:
        r3 = lambda v: float(f'{v:.3g}')
        grads = {'status': 'stale' if gstale else 'measured', 'stale': gstale,
                 'source': 'tests/test_full_size_parity.py + tests/test_e1_full_gpu.py -> profiles/' + f"{ROUND}_parity_gradients_{'f16' if mode == 'default' else mode}.json"}
        tie = {}
        if 'generator' in g:
            tie['generator'] = r3(g['generator']['tie_masked_worst'][1])
        if 'discriminator' in g:
            tie['discriminator_G_loss'] = r3(g['discriminator']['tie_masked_worst_G_loss'][1])
            tie['discriminator_D_loss'] = r3(g['discriminator']['tie_masked_worst_D_loss'][1])
        for k in ('vgg19', 'vggface'):
            if k in g:
                tie[k + '_d_image'] = r3(g[k]['tie_masked_d_fake'])
        grads['tie_masked_worst_rel_l2'] = tie
        # the plain figures (against the true-ReLU / true-sign oracle) beside them: a pre-activation within rounding distance of 0 flips its ReLU
        # between two correct implementations, which the tie-masked comparison removes and these keep (fp32 CPU oracle vs fp64: generator 2.7e-4)
        untied = {}
        if 'generator' in g:
            untied['generator'] = r3(g['generator']['untied_worst'])
            untied['generator_fp32_oracle_tie_floor_vs_fp64'] = r3(g['generator']['fp32_oracle_tie_floor_vs_fp64'])
        if 'discriminator' in g and 'untied_worst_D_loss' in g['discriminator']:
            untied['discriminator_G_loss'] = r3(g['discriminator']['untied_worst_G_loss'])
            untied['discriminator_D_loss'] = r3(g['discriminator']['untied_worst_D_loss'])
        for k in ('vgg19', 'vggface'):
            if k in g:
                untied[k + '_d_image'] = r3(g[k]['untied_d_fake'])
        grads['untied_worst_rel_l2'] = untied
        grads['within_1e-3'] = bool(tie) and all(v < 1e-3 for v in tie.values())
        if 'identity_encoder' in g:
            e = g['identity_encoder']
            grads['identity_encoder'] = {'all_gradients_rel_l2': r3(e['all_gradients_rel']), 'all_gradients_cosine': r3(e['all_gradients_cosine']),
                                         'stock_fp32_layers_all_gradients_rel_l2': r3(e['stock_fp32_layers_vs_fp64']['all_gradients_rel']),
                                         'note': 'train-mode BatchNorm at random initialisation is chaotic: the stock fp32 layers (the reference\'s own '
                                                 'arithmetic class) are this far from fp64 themselves; no 1e-3 claim is made for these gradients in any mode'}
        if 'identity_encoder_tie_masked' in g:
            e = g['identity_encoder_tie_masked']
            grads['identity_encoder_tie_masked'] = {
                'all_gradients_rel_l2': r3(e['all_gradients_rel']), 'all_gradients_cosine': float(f"{e['all_gradients_cosine']:.7g}"),
                'embeds': r3(e['embeds']), 'per_frame_logits': r3(e['per_frame_logits']),
                'stock_fp32_layers_all_gradients_rel_l2': r3(e['stock_fp32_layers_vs_fp64']['all_gradients_rel']),
                'note': 'the well-conditioned full-depth check: the same 64 frames and train-mode BatchNorm, the fp64 stock layers evaluated on the HIP '
                        'path\'s own ReLU patterns and max-pool argmax (as the G / D / VGG figures are); gate 1e-2 on the all-parameter gradient '
                        '(tests/test_e1_full_gpu.py)'}
        if 'pose_encoder_tie_masked' in g:
            e = g['pose_encoder_tie_masked']
            grads['pose_encoder_tie_masked'] = {
                'all_gradients_rel_l2': r3(e['all_gradients_rel']), 'all_gradients_cosine': float(f"{e['all_gradients_cosine']:.9g}"),
                'worst_parameter_tensor': [e['worst_parameter_tensor'][0], r3(e['worst_parameter_tensor'][1])], 'pose_vector': r3(e['pose_vector']),
                'plain_all_gradients_rel_l2': r3(g['pose_encoder']['all_gradients_rel']) if 'pose_encoder' in g else None,
                'stock_fp32_layers_all_gradients_rel_l2': r3(e['stock_fp32_layers_vs_fp64']['all_gradients_rel']),
                'note': 'MobileNetV2 pose encoder, 8 frames of 256 x 256, train-mode BatchNorm, bf16x3 contractions; the fp64 stock layers evaluated on the '
                        'HIP path\'s own 35 ReLU6 branch patterns; gate 1e-3 on the all-parameter gradient (tests/test_e2_full_gpu.py)'}
        if gstale:
            grads['stale_reason'] = gwhy
        out['gradients'] = grads
    out['gates_met'] = {'forward_quantities_and_losses_1e-3': out['within_1e-3'] and not stale,
                        'tie_masked_parameter_gradients_1e-3 (G, D, VGG)': bool(out['gradients'].get('within_1e-3')) and not out['gradients'].get('stale', True)}
    if workload != 'metatrain_step':
        out['note'] = 'figures of the meta-training configuration (this workload shares its generator, discriminator and criterions; its pose encoder runs without autograd)'
    return out


def self_launch(n):
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # stdout of the job is filtered to the JSON line(s): libraries of the ranks (gloo's "[Gloo] Rank 0 is connected ..." banner, launcher
    # notices) also write there, and the driver reads stdout as ONE JSON line
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        if line.lstrip().startswith('{'):
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc:
        raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=8, help='per-GPU batch')
    ap.add_argument('--image_size', type=int, default=256)
    ap.add_argument('--prec', default=os.environ.get('LP_PREC', 'f16'), choices=['bf16', 'bf16x3', 'f16'],
                    help='MFMA operand mode: f16 (default; 1 MFMA per MAC, outputs 1.7e-4 from the fp32 CPU path), bf16x3 (strict: 3 MFMAs, 2.5e-6), bf16')
    ap.add_argument('--cpu-baseline-full', action='store_true', help='BASELINE.md section 3 protocol (>= 3 warm-up + >= 10 timed CPU steps, minutes)')
    ap.add_argument('--generator', default='vector_pose_unsupervised_segmentation_noBottleneck', choices=['vector_pose_unsupervised_segmentation_noBottleneck', 'FSTH_plus'],
                    help='generator plugin of --workload generator (FSTH_plus with --image_size 512 --batch 4 = BASELINE configs[4])')
    ap.add_argument('--workload', default=None, choices=['finetune_step', 'metatrain_step', 'generator'],
                    help='default: metatrain_step for EVERY --gpus value (configs[2], default.yaml: the configuration that is trained data-parallel; one '
                         'workload so that the 1/2/4/8-GPU values form a scaling curve).  finetune_step = BASELINE configs[1] (single GPU in the '
                         'reference); the default N = 1 run also reports it under "finetune_step"')
    ap.add_argument('--padding', default='zero', choices=['zero', 'reflection'], help='--gen_padding / --dis_padding of the timed step (the shipped configs and the headline line: zero)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-only', action='store_true', help='only the CPU baseline leg (no GPU work): prints its JSON object')
    ap.add_argument('--no-also', action='store_true', help='skip the side measurements (strict bf16x3 mode, fine-tuning step)')
    ap.add_argument('--no-drive', action='store_true', help='skip the drive.py frame-loop measurement (PMC passes: keeps the launch population to the step)')
    ap.add_argument('--eager', action='store_true', help='do not capture the step into hipGraphs')
    ap.add_argument('--shapes', default=None, help='write the per-shape conv / wgrad timing table of the instrumented steps (CSV)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL; gloo only for functional tests)')
    ap.add_argument('--cpu-worker', default=None, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.cpu_worker:
        return cpu_worker(a.cpu_worker)
    if a.cpu_baseline_only:
        wl = a.workload or 'metatrain_step'
        print(json.dumps(cpu_baseline(make_args(a.image_size, a.batch, 'cpu', 1, 0, a.prec, finetune=wl != 'metatrain_step'), full=a.cpu_baseline_full, workload=wl)))
        return

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU under torch.distributed.run, rendezvous on
        # 127.0.0.1, a free port) with the same command line; rank 0 of the job prints the JSON line on the inherited stdout
        return self_launch(a.gpus)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the HIP path)')
    dev_index = local_rank % torch.cuda.device_count()      # (modulo only matters for single-GPU functional tests of the N > 1 path)
    torch.cuda.set_device(dev_index)
    device = f'cuda:{dev_index}'
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend=a.backend, init_method='env://')
    assert world == a.gpus, f'--gpus {a.gpus} but WORLD_SIZE={world}'

    default_run = a.workload is None
    if a.workload is None:
        a.workload = 'metatrain_step'
    finetune = a.workload != 'metatrain_step'
    args = make_args(a.image_size, a.batch, device, world, rank, a.prec, finetune=finetune)
    args.gen_padding = args.dis_padding = a.padding
    args.generator = a.generator
    if a.generator == 'FSTH_plus':
        args.pose_embedding_size = 136          # 68 landmarks x 2 (generators/FSTH_plus.py:129-139)
    tm, opt_G, opt_D, holycow = build(args)
    if world > 1:
        from latent_pose_reenactment_amd.parallel import GradReducer
        tm.reducer = GradReducer(tm, finetune=finetune, optimizer_G=opt_G, optimizer_D=opt_D, max_batch=a.batch)
    data, target = synthetic_batch(args, a.batch, seed=123 + rank)

    from latent_pose_reenactment_amd import hipops

    def gen_only_step():
        dd = {'pose_embedding': pose_static, 'dec_keypoints': kp_static}
        tm.generator(dd)
        (dd['fake_rgbs'].mean() + dd['fake_segm'].mean()).backward()

    if a.workload == 'generator':
        pose_static = torch.randn(a.batch, args.pose_embedding_size, device=device)
        kp_static = torch.rand(a.batch, 1, 136, device=device)
        step = gen_only_step
    else:
        def eager_step():
            holycow.train_step(tm, data, target, opt_G, opt_D, args)
        step = eager_step
        mode = 'eager'
        if not a.eager:
            try:
                step = holycow.GraphedTrainStep(tm, opt_G, opt_D, args, data, target, warmup_steps=max(a.warmup, 2))
                mode = 'hipgraph'
            except Exception as ex:        # keep the bench alive: fall back to the eager loop and say so
                print(f'[bench] hipGraph capture failed ({ex!r}); running eagerly', file=sys.stderr)
                step = eager_step

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    # Live per-kernel timing for the roofline entry: HIP events recorded on the launch stream around every conv launch.
    # Graph replays cannot carry per-kernel events, so the instrumented steps run eagerly right after the timed region
    # (same process, same buffers, same kernels); they are not part of `value`.
    # They run on ONE stream (LP_OVERLAP=0): beside a concurrent branch (streams.py) a kernel's event span would include the time it shares
    # the CUs with another stream's kernels, which is not what a per-kernel roofline states.
    hipops.PROFILE = []
    inst = eager_step if a.workload != 'generator' else step
    keep_overlap = os.environ.get('LP_OVERLAP')
    os.environ['LP_OVERLAP'] = '0'
    try:
        for _ in range(2):
            inst()
        torch.cuda.synchronize()
    finally:
        if keep_overlap is None:
            os.environ.pop('LP_OVERLAP', None)
        else:
            os.environ['LP_OVERLAP'] = keep_overlap
    prof, hipops.PROFILE = hipops.PROFILE, None
    replicas = None
    if world > 1 and a.workload != 'generator':
        # data-parallel invariant (apex Reducer semantics, train.py:196-200): after any number of steps every rank holds bit-identical parameters
        # (same start by broadcast, same averaged gradients, same optimizer arithmetic).  Checked on the parameters' BIT patterns: two int64
        # checksums per rank, MIN and MAX over ranks must agree.
        with torch.no_grad():
            groups = {'generator': list(tm.generator.parameters()), 'embedder': list(tm.embedder.parameters()),
                      'discriminator.embed': [p_ for k_, p_ in tm.discriminator.named_parameters() if k_.startswith('embed.')],
                      'discriminator.rest': [p_ for k_, p_ in tm.discriminator.named_parameters() if not k_.startswith('embed.')]}
            sums, total = [], 0
            for ps in groups.values():
                flat = torch.cat([p_.detach().reshape(-1) for p_ in ps])
                bits = flat.view(torch.int32).to(torch.int64)
                sums += [bits.sum(), (bits * (torch.arange(bits.numel(), device=device, dtype=torch.int64) % 65521 + 1)).sum()]
                total += flat.numel()
            cs = torch.stack(sums)
            lo, hi = cs.clone(), cs.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            same = (lo == hi).view(-1, 2).all(dim=1).tolist()
            replicas = {'bit_identical_parameters': bool(all(same)), 'parameters': int(total),
                        'by_group': {k_: bool(v_) for k_, v_ in zip(groups, same)},
                        'after_steps': a.warmup + a.steps + 2,
                        'note': 'checked after the warm-up, timed and instrumented steps, BEFORE the exchange-free single-GPU leg (which lets the ranks drift apart on purpose)'}

    comm = None
    if world > 1 and a.workload != 'generator' and getattr(tm, 'reducer', None) is not None:
        # diagnosis of the gradient exchange: a few more steps of the SAME step function with events around every
        # all-reduce issue and around the wait of its consumer -- per-bucket bytes, how long each collective had to finish behind other work,
        # and how long the compute stream still stood in the wait (exposed_comm_ms_per_step).  Outside the timed region.
        try:
            nd = 5
            tm.reducer.enable_diag()
            for _ in range(nd):
                step()
            sync()
            comm = tm.reducer.diag_summary(nd)
            tm.reducer.diag = None
            for b_ in [tm.reducer.g_bucket, tm.reducer.d_bucket] + list(tm.reducer.g_parts.values()):
                b_.diag = None
            if comm is not None:
                comm['backend'] = a.backend
                comm['reduce_op'] = 'AVG (inside the collective)' if a.backend == 'nccl' else 'SUM + div_'
        except Exception as ex:
            comm = {'error': repr(ex)}

    solo = None
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
        if a.workload != 'generator':
            # the N = 1 point of THIS workload measured inside the same job (a cross-check of the driver's separate --gpus 1 run):
            # every rank repeats the timed loop with the gradient exchange removed -- one GPU's throughput on the same step and batch
            red, tm.reducer = tm.reducer, None
            nsolo = min(a.steps, 50)
            t_local, err = 0.0, None
            try:      # no collective inside: a rank that fails here must not leave the others waiting in one
                step1 = eager_step if a.eager else holycow.GraphedTrainStep(tm, opt_G, opt_D, args, data, target, warmup_steps=2)
                for _ in range(3):
                    step1()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(nsolo):
                    step1()
                torch.cuda.synchronize()
                t_local = time.perf_counter() - t1
            except Exception as ex:
                err = repr(ex)
            t = torch.tensor([t_local, 0.0 if err is None else 1.0], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if t[1].item() > 0:
                solo = {'error': err or 'failed on another rank'}
            else:
                solo = {'value': round(a.batch * nsolo / t[0].item(), 3), 'unit': 'images/s', 'n_gpus': 1, 'steps': nsolo,
                        'ms_per_step': round(t[0].item() / nsolo * 1e3, 3),
                        'note': 'same workload and per-GPU batch with the gradient exchange removed, timed on every rank at once (max over ranks): '
                                'the single-GPU point of this curve'}
            tm.reducer = red

    # live roofline of the dominant kernel family (HIP events recorded on the launch stream inside the timed region)
    agg = {}
    shapes = {}
    agg_bytes = {}
    agg_rw = {}
    agg_mm = {}          # kind -> {MFMAs per MAC: [flops, seconds, launches]}
    for kind, flops, e0, e1, tag, nbytes, rw, mm in prof:
        qm = agg_mm.setdefault(kind, {}).setdefault(mm, [0.0, 0.0, 0])
        qm[0] += flops; qm[1] += e0.elapsed_time(e1) * 1e-3; qm[2] += 1
        if rw is not None:
            q = agg_rw.setdefault(kind, [0.0, 0.0, 0])
            q[0] += rw[0]; q[1] += rw[1]; q[2] += 1
        sd = shapes.setdefault((kind, tag), [0.0, 0.0, 0])
        sd[0] += flops; sd[1] += e0.elapsed_time(e1) * 1e-3; sd[2] += 1
        d = agg.setdefault(kind, [0.0, 0.0, 0])
        d[0] += flops; d[1] += e0.elapsed_time(e1) * 1e-3; d[2] += 1
        agg_bytes[kind] = agg_bytes.get(kind, 0.0) + nbytes
    HBM_KINDS = ('conv1x1', 'wgrad1x1', 'gconv', 'gconv_wgrad')      # the embedder's contractions: bound by their activation traffic, not by MFMA
    HBM_PEAK_GBS = 8000.0
    roof = None
    extra = {}
    for kind, (fl, sec, cnt) in agg.items():
        ach = fl / sec / 1e12 if sec > 0 else 0.0
        entry = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                 'frac': round(ach / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': None, 'kernel': kind,
                 'launches': cnt, 'avg_launch_us': round(sec / max(cnt, 1) * 1e6, 1),
                 'algorithmic_gflop_per_launch': round(fl / max(cnt, 1) / 1e9, 3),
                 'traffic_note': None}
        # bf16x3 launches execute 3 MFMAs per algorithmic MAC (hi*hi + hi*lo + lo*hi).  `frac` prices the ALGORITHMIC flops; the matrix
        # pipe's own work -- what the kernel is bound by -- is printed beside it, with the family split by operand mode (round 6: the default
        # assignment runs the generator and most critic launches of this family with bf16x3 operands to meet the 1e-3 gradient gate)
        by_mode = {}
        work = 0.0
        for mm_, (fl_, sec_, cnt_) in sorted(agg_mm.get(kind, {}).items()):
            work += fl_ * mm_
            by_mode['bf16x3 (3 MFMAs per MAC)' if mm_ == 3 else 'one MFMA per MAC (f16 / bf16)'] = {
                'launches': cnt_, 'seconds_share': round(sec_ / sec, 3) if sec > 0 else None,
                'algorithmic_tflops': round(fl_ / sec_ / 1e12, 1) if sec_ > 0 else None,
                'frac_algorithmic': round(fl_ / sec_ / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4) if sec_ > 0 else None,
                'frac_mfma_work': round(fl_ * mm_ / sec_ / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4) if sec_ > 0 else None}
        entry['mfma_per_algorithmic_flop'] = round(work / fl, 3) if fl > 0 else 1
        entry['mfma_work_tflops'] = round(work / sec / 1e12, 2) if sec > 0 else 0.0
        entry['frac_mfma_work'] = round(work / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4) if sec > 0 else 0.0
        entry['by_operand_mode'] = by_mode
        if kind in HBM_KINDS and sec > 0:
            gbs = agg_bytes.get(kind, 0.0) / sec / 1e9
            entry.update({'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4),
                          'algorithmic_mb_per_launch': round(agg_bytes.get(kind, 0.0) / max(cnt, 1) / 1e6, 2), 'algorithmic_tflops': round(ach, 1),
                          'traffic_note': 'algorithmic bytes: operand planes read once + weights + the fp32 (and plane) outputs written once; '
                                          'priced against the 8 TB/s HBM3E peak (6.3 TB/s is what a streaming copy reaches on this chip)'})
            for k_ in ('mfma_per_algorithmic_flop', 'mfma_work_tflops', 'algorithmic_gflop_per_launch', 'frac_mfma_work', 'by_operand_mode'):
                entry.pop(k_, None)
        if kind == 'conv_igemm':
            # HBM bytes per launch of this kernel family from the committed PMC passes: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs
            # of THIS workload (the meta-training step), counter collection restricted to the 3x3 kernels (scripts/r06_artifacts.sh; FETCH_SIZE
            # doubled per MI355X_MICROARCH.md).  The file carries the source stamp of the tree it was measured on: a stale file is SAID to be stale.
            pm, pstale, pwhy = _load_profile(f'{ROUND}_pmc_conv3x3_metatrain.json')
            if pm is not None:
                entry['traffic'] = pm.get('hbm_bytes_per_launch')
                entry['traffic_stale'] = pstale
                # read and write separately, with the algorithmic bytes of the same family beside them (mean over this process's instrumented launches)
                rw = agg_rw.get(kind)
                entry['traffic_read_write'] = {'hbm_read_bytes_per_launch': pm.get('hbm_read_bytes_per_launch'), 'hbm_write_bytes_per_launch': pm.get('hbm_write_bytes_per_launch'),
                                               'algorithmic_read_bytes_per_launch': None if not rw else int(rw[0] / rw[2]),
                                               'algorithmic_write_bytes_per_launch': None if not rw else int(rw[1] / rw[2]),
                                               'family': 'dense 3x3 forward / dgrad: conv_pipe_kernel + conv_dma_kernel<3> (+ <2>: the phase forms of the x2-upsampled convs) without the grouped (identity-encoder) instantiations',
                                               'read_amplification': None if not (rw and pm.get('hbm_read_bytes_per_launch')) else round(pm['hbm_read_bytes_per_launch'] / (rw[0] / rw[2]), 2)}
                entry['traffic_note'] = ('mean HBM bytes per DENSE 3x3 conv launch (conv_pipe_kernel + conv_dma_kernel<3>, grouped instantiations excluded) over the launch population of the '
                                         'meta-training step (warm-up, capture and replays of `bench.py --steps 2 --warmup 1`): profiles/' + ROUND + '_pmc_conv3x3_metatrain.json, '
                                         'separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per MI355X_MICROARCH.md; '
                                         f"MFMA busy fraction from SQ_VALU_MFMA_BUSY_CYCLES: {pm.get('mfma_busy_fraction')}" + (f' -- STALE: {pwhy}' if pstale else ''))
                entry['mfma_busy_fraction_pmc'] = pm.get('mfma_busy_fraction')
            else:
                entry['traffic_note'] = 'no PMC summary of this round under profiles/ (' + pwhy + ')'
            entry['algorithmic_bytes_note'] = ('f16 operands: 2 B per input activation (once per N tile), 2 B per weight, 4 B per fp32 output '
                                               '(+ 2 B when the epilogue also emits the consumer planes; planes-only outputs: 2 B)')
            entry['frac_note'] = ('frac / achieved: LIVE -- HIP events on the launch stream around every launch of the family in two eager one-stream steps of this '
                                  'process (each event pair adds a few microseconds to a ~35 us launch); in_graph: the same family inside the captured step '
                                  '(rocprofv3 kernel trace of a graph replay x the shape list, scripts/in_graph_conv.py) -- the figure without eager launch gaps')
            ig, istale, iwhy = _load_profile(f'{ROUND}_conv3x3_in_graph.json')
            if ig is not None:      # the same family INSIDE the graph replay (rocprofv3 kernel trace of the captured step: no eager launch gaps)
                entry['in_graph'] = dict(ig, stale=istale, **({'stale_reason': iwhy} if istale else {}))
                entry['frac_in_graph'] = ig.get('frac')
        if kind == 'conv_igemm':
            entry['kernel'] = 'conv_pipe_kernel / conv_dma_kernel<3> (lp_conv16_fwd: forward and data-gradient 3x3 convs)'
            roof = entry
        else:
            extra['roofline_' + kind] = entry

    if rank == 0 and a.shapes:
        with open(a.shapes, 'w') as f:
            f.write('kind,N,H,W,Cin,Cout,ksize,upsample,prologue,launches,total_us,avg_us,TFLOPs\n')
            for (kind, tag), (fl, sec, cnt) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                f.write(','.join([kind] + [str(v) for v in (tag or ())] + [str(cnt), f'{sec * 1e6:.1f}', f'{sec / cnt * 1e6:.1f}',
                                                                          f'{fl / sec / 1e12:.1f}']) + '\n')
    if rank == 0:
        imgs = a.batch * world * a.steps
        out = {
            'metric': 'train-step images/sec at 256x256 bs=8' if a.workload != 'generator' else 'generator fwd+bwd images/sec at 256x256 bs=8',
            'value': round(imgs / dt, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'bf16x3': 'bf16x3 (hi+lo split bf16 MFMA operands, 3 MFMAs per MAC, fp32 accumulate)', 'bf16': 'bf16 (MFMA operands, fp32 accumulate)',
                      'f16': 'mixed 16-bit MFMA operands, fp32 accumulate, fp32 activations / weights / optimizer state -- default assignment (round 6, chosen so that '
                             'every forward quantity AND every parameter gradient is within 1e-3 of the fp32 CPU path): '
                             + (assignment_description(tm) if a.workload != 'generator' else f'generator: {"bf16x3" if tm.generator.prec == 1 else "f16"}')
                             + '; bf16x3 = hi + lo split bf16 operands, 3 MFMAs per MAC; f16 = IEEE fp16 operands incl. power-of-two scaled gradient operands'
                             + (encoder_modes(tm) if a.workload == 'metatrain_step' else '')}[a.prec],
            'data': 'synthetic VoxCeleb2-shaped batch, random-init weights (VGG weights seeded He-normal)',
            'config': {'workload': {'finetune_step': 'finetuning-base.yaml step (configs[1]): G, D, VGG19/VGGFace criterions, RAdam, EMA, spectral '
                                                     'norm and the MobileNetV2 pose encoder on hand-written gfx950 kernels',
                                    'metatrain_step': 'default.yaml meta-training step (configs[2]): ResNeXt50 identity encoder over 8 frames + '
                                                      'MobileNetV2 pose encoder (both trained, forward + backward on hand-written gfx950 kernels), G, D with '
                                                      'the 98000x512 label embedding, VGG19/VGGFace/featmat/adversarial/dis_embed/dice criterions, Adam, EMA',
                                    'generator': 'generator forward+backward only (HIP kernels)'}[a.workload],
                       'image_size': a.image_size, 'per_gpu_batch': a.batch, 'global_batch': a.batch * world,
                       'parallelism': f'dp{world}', 'precision_mode': a.prec if a.prec != 'f16' else 'default assignment (LP_PREC=f16)', 'padding': a.padding,
                       'launch_mode': mode if a.workload != 'generator' else 'eager',
                       'streams': stream_config(a.workload)},
            'roofline': roof,
        }
        # whole-step MFMA utilisation: the number the 0.60 target of BASELINE.json is about (useful dense FLOPs of the step / time / peak)
        tf = step_algorithmic_tflop(a.workload, a.batch, args.n_frames_for_encoder, a.image_size, a.generator) * world
        out['step_mfma_utilisation'] = {'algorithmic_tflop_per_step': round(tf, 3), 'achieved_tflops': round(tf / (dt / a.steps), 1),
                                        'frac_of_bf16_peak': round(tf / (dt / a.steps) / (MFMA_BF16_PEAK_TFLOPS * world), 4),
                                        'note': 'useful dense 2*MAC of one step as executed (step_algorithmic_tflop) / ms_per_step / (2.5 PF x n_gpus)'}
        out.update(extra)
        if a.workload != 'generator':
            out['parity'] = measured_parity('default' if a.prec == 'f16' and not os.environ.get('LP_PREC_E') else a.prec, a.workload)
        if solo is not None:
            out['single_gpu_same_workload'] = solo
        if replicas is not None:
            out['replicas'] = replicas
        if comm is not None:
            out['gradient_exchange'] = comm
        if world == 1 and not a.no_drive:
            try:
                out['drive'] = drive_fps(args)
                out['drive']['batch_8'] = {k: v for k, v in drive_fps(args, frames=30, batch=8).items() if k != 'note'}
            except Exception as ex:
                out['drive'] = {'error': repr(ex)}
        def child(extra_args, timeout=600):
            import subprocess
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-also'] + extra_args,
                               capture_output=True, text=True, timeout=timeout)
            return json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
        if world == 1 and a.prec == 'f16' and a.workload in ('finetune_step', 'metatrain_step') and not a.no_also:
            # the strict-parity mode (bf16x3: 3 MFMAs per MAC, fp32-class results) of the SAME workload, measured by the same script in a
            # child process.  Parity status of the two modes: the f16 headline meets the 1e-3 OUTPUT gate of BASELINE.json; the full
            # SURVEY 8(d) gate (outputs, losses AND every parameter gradient within 1e-3) is met in bf16x3 only.
            try:
                j = child(['--prec', 'bf16x3', '--workload', a.workload])
                out['strict_mode_bf16x3'] = {'value': j['value'], 'unit': j['unit'], 'ms_per_step': j['ms_per_step'],
                                             'roofline_frac': (j.get('roofline') or {}).get('frac'),
                                             'parity': measured_parity('bf16x3', a.workload)}
            except Exception as ex:
                out['strict_mode_bf16x3'] = {'error': repr(ex)}
        if world == 1 and a.prec == 'f16' and a.workload in ('finetune_step', 'metatrain_step') and not a.no_also \
                and not (os.environ.get('LP_PREC_G') or os.environ.get('LP_D_DPASS_PREC')):
            # round 5's assignment (fp16 operands in the generator and in the critic's D-side passes) of the same workload: faster, meets the 1e-3 gate
            # on every forward quantity and loss but NOT on the parameter gradients (generator 1.0 - 1.7e-3, critic <= 9.3e-3 tie-masked) -- an option
            # (LP_PREC_G=f16 LP_D_DPASS_PREC=f16), not the headline
            try:
                import subprocess
                r = subprocess.run([sys.executable, os.path.abspath(__file__), '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-also', '--no-drive',
                                    '--workload', a.workload], capture_output=True, text=True, timeout=600,
                                   env=dict(os.environ, LP_PREC_G='f16', LP_D_DPASS_PREC='f16'))
                j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
                out['fp16_generator_and_critic_option'] = {
                    'value': j['value'], 'unit': j['unit'], 'ms_per_step': j['ms_per_step'], 'env': 'LP_PREC_G=f16 LP_D_DPASS_PREC=f16',
                    'gates_met': {'forward_quantities_and_losses_1e-3': True, 'tie_masked_parameter_gradients_1e-3 (G, D, VGG)': False},
                    'note': 'round 5\'s headline assignment; its gradient figures: profiles/r05_parity_gradients_f16.json and the `generator_f16_operands` / '
                            '`discriminator_f16_operands` entries of profiles/' + ROUND + '_parity_gradients_f16.json'}
            except Exception as ex:
                out['fp16_generator_and_critic_option'] = {'error': repr(ex)}
        if world == 1 and default_run and not a.no_also:
            # BASELINE configs[1] (fine-tuning step, single GPU in the reference) beside the scaling workload
            try:
                j = child(['--prec', a.prec, '--workload', 'finetune_step'])
                out['finetune_step'] = {k: j.get(k) for k in ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'roofline', 'step_mfma_utilisation')}
                out['finetune_step']['config'] = j.get('config')
            except Exception as ex:
                out['finetune_step'] = {'error': repr(ex)}
        if world == 1 and not a.no_cpu_baseline and a.workload != 'generator':
            try:
                out['cpu_baseline'] = cpu_baseline(args, full=a.cpu_baseline_full, workload=a.workload)
            except Exception as ex:      # never lose the GPU line because of the baseline leg
                out['cpu_baseline'] = {'error': repr(ex)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
