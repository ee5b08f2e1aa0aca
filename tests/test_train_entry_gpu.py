"""train.py's own entry point (reference train.py:92-310 flow: config merge, plugin loading, epoch loop, checkpoint) on the
synthetic dataloader, eager and with ``--hip_graph``: the graph-replayed run_epoch must leave the same weights as the eager loop
(same optimizer-step sequence: the first iterations run eagerly, then the captured step is replayed on every new batch), the
iteration counter must advance (checkpoint names model_{iteration:08}.pth), and a stale weight-pack cache must not survive an
optimizer step."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'latent_pose_reenactment_amd')
pytestmark = pytest.mark.gpu

COMMON = ['--generator', 'vector_pose_unsupervised_segmentation_noBottleneck', '--embedder', 'unsupervised_pose_separate_embResNeXt_segmentation',
          '--discriminator', 'no_landmarks', '--criterions', 'adversarial,featmat,dis_embed,dice', '--runner', 'holycow',
          '--dataloader', 'synthetic_voxceleb2', '--image_size', '128', '--num_channels', '4', '--max_num_channels', '16',
          '--embed_channels', '8', '--pose_embedding_size', '4', '--num_labels', '50', '--dis_num_blocks', '5', '--batch_size', '4',
          '--synthetic_dataset_len', '32', '--n_frames_for_encoder', '2', '--num_epochs', '1', '--num_gpus', '1']
# (128 px, 4 samples x 2 encoder frames = 8 frames: the smallest geometry the hand-written encoders accept -- the product has no other backend.)
# (train mode: spectral-norm power iterations, BatchNorm batch statistics and the pose encoder's dropout are live; torch's graph-safe
#  Philox generator hands a replayed step the same offsets the eager step would have used, so the two runs see the same dropout masks)


def run_train(tmp, name, extra):
    env = dict(os.environ, LP_PREC='bf16x3')
    cmd = [sys.executable, os.path.join(PKG, 'train.py')] + COMMON + ['--experiments_dir', str(tmp), '--experiment_name', name] + extra
    r = subprocess.run(cmd, cwd=PKG, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ckpts = sorted(os.listdir(os.path.join(str(tmp), name, 'checkpoints')))
    line = [l for l in r.stdout.splitlines() if 'Epoch 0 (iteration' in l][-1]
    losses = {m.group(1): float(m.group(2)) for m in re.finditer(r'(Loss_\w+) ([-+.\deE]+) \(avg', line)}
    return ckpts, torch.load(os.path.join(str(tmp), name, 'checkpoints', ckpts[-1]), map_location='cpu', weights_only=False), losses


def _compare(a_ckpt, a_loss, b_ckpt, b_loss):
    lerr = max(abs(b_loss[k] - a_loss[k]) / max(abs(a_loss[k]), 1e-2) for k in a_loss)
    serr = 0.0
    for part in ('generator', 'discriminator', 'embedder'):
        # parameters only (BatchNorm running variances of 1x1 feature maps amplify any noise), as one vector per module
        a = torch.cat([v.double().reshape(-1) for k, v in a_ckpt[part].items() if v.dtype == torch.float32 and 'running_' not in k])
        b = torch.cat([v.double().reshape(-1) for k, v in b_ckpt[part].items() if v.dtype == torch.float32 and 'running_' not in k])
        serr = max(serr, ((a - b).norm() / a.norm()).item())
    return lerr, serr


def test_run_epoch_with_hip_graph_equals_eager(tmp_path):
    """Two EAGER runs of this loop already differ: float atomics of the crop-and-resize backward add rounding noise, BatchNorm over a
    handful of values and Adam (which moves an element whose true gradient is ~0 by +-lr) amplify it.  So the replayed loop is
    held to the eager loop's own run-to-run spread (measured here by a second eager run), on quantities that are smooth in the
    weights: the losses of the LAST iteration -- off at once if a replay saw a stale batch or lost an update -- and each module's
    parameter vector."""
    names_e, eager, loss_e = run_train(tmp_path, 'eager', ['--log_frequency_loss', '1'])
    _, eager2, loss_e2 = run_train(tmp_path, 'eager2', ['--log_frequency_loss', '1'])
    names_g, graph, loss_g = run_train(tmp_path, 'graph', ['--hip_graph', '--log_frequency_loss', '1'])
    assert names_e == names_g == ['model_00000008.pth'], (names_e, names_g)      # 32 samples / batch 4 = 8 iterations
    assert eager['args'].iteration == 8 and graph['args'].iteration == 8
    assert set(loss_e) == set(loss_g) and len(loss_e) >= 5, (loss_e, loss_g)
    noise_l, noise_s = _compare(eager, loss_e, eager2, loss_e2)
    # the replayed run against the NEARER of the two eager runs (one sample of the eager-vs-eager spread is a noisy yardstick: a single
    # comparison at 3x that sample fails a few percent of the time on this chaotic toy configuration)
    lerr, serr = (min(x) for x in zip(_compare(eager, loss_e, graph, loss_g), _compare(eager2, loss_e2, graph, loss_g)))
    print(f'[entry] --hip_graph vs eager after 8 iterations of run_epoch: last-iteration losses {lerr:.2e} (eager vs eager {noise_l:.2e}); '
          f'parameter vectors {serr:.2e} (eager vs eager {noise_s:.2e})')
    assert lerr < max(3 * noise_l, 2e-3), (lerr, noise_l, loss_e, loss_g)
    assert serr < max(3 * noise_s, 1e-3), (serr, noise_s)


def test_inference_pack_cache_follows_optimizer_and_ema_updates():
    """eval forward (packs cached) -> fused optimizer step + EMA (raw-pointer updates) -> eval forward must use fresh packs"""
    sys.path.insert(0, PKG)
    from latent_pose_reenactment_amd.nn import Generator
    from latent_pose_reenactment_amd.optim import FusedAdam
    torch.manual_seed(0)
    G = Generator('zero', 3, 4, 8, 32, 16, 8, 'in', 4, 2, 32, prec=1).cuda().eval()
    d = {'embeds': torch.randn(2, 16).cuda(), 'pose_embedding': torch.randn(2, 8).cuda()}
    with torch.no_grad():
        G(dict(d))
        assert G.__dict__.get('_pack_cache') is not None
    opt = FusedAdam(G.parameters(), lr=1e-2, betas=(0.0, 0.999), eps=1e-5)
    opt.zero_grad()
    for p in G.parameters():
        p.grad.normal_()
    opt.step()
    with torch.no_grad():
        a = dict(d); G(a)
        G.__dict__['_pack_cache'] = None
        b = dict(d); G(b)
    assert torch.equal(a['fake_rgbs'], b['fake_rgbs'])
