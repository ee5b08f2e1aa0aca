"""train.py's own entry point (reference train.py:92-310 flow: config merge, plugin loading, epoch loop, checkpoint) on the
synthetic dataloader, eager and with ``--hip_graph``: the graph-replayed run_epoch must leave the same weights as the eager loop
(same optimizer-step sequence: the first iterations run eagerly, then the captured step is replayed on every new batch), the
iteration counter must advance (checkpoint names model_{iteration:08}.pth), and a stale weight-pack cache must not survive an
optimizer step."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'latent_pose_reenactment_amd')
pytestmark = pytest.mark.gpu

COMMON = ['--generator', 'vector_pose_unsupervised_segmentation_noBottleneck', '--embedder', 'unsupervised_pose_separate_embResNeXt_segmentation',
          '--discriminator', 'no_landmarks', '--criterions', 'adversarial,featmat,dis_embed,dice', '--runner', 'holycow',
          '--dataloader', 'synthetic_voxceleb2', '--image_size', '32', '--num_channels', '4', '--max_num_channels', '16',
          '--embed_channels', '8', '--pose_embedding_size', '4', '--num_labels', '50', '--dis_num_blocks', '5', '--batch_size', '2',
          '--synthetic_dataset_len', '16', '--n_frames_for_encoder', '2', '--num_epochs', '1', '--num_gpus', '1',
          '--set_eval_mode_in_train', '--log_frequency_loss', '4']      # eval mode: no dropout / batch statistics -> bit-comparable runs


def run_train(tmp, name, extra):
    env = dict(os.environ, LP_PREC='bf16x3')
    cmd = [sys.executable, os.path.join(PKG, 'train.py')] + COMMON + ['--experiments_dir', str(tmp), '--experiment_name', name] + extra
    r = subprocess.run(cmd, cwd=PKG, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ckpts = sorted(os.listdir(os.path.join(str(tmp), name, 'checkpoints')))
    return ckpts, torch.load(os.path.join(str(tmp), name, 'checkpoints', ckpts[-1]), map_location='cpu', weights_only=False)


def test_run_epoch_with_hip_graph_equals_eager(tmp_path):
    names_e, eager = run_train(tmp_path, 'eager', [])
    names_g, graph = run_train(tmp_path, 'graph', ['--hip_graph'])
    assert names_e == names_g == ['model_00000008.pth'], (names_e, names_g)      # 16 samples / batch 2 = 8 iterations, counted on rank 0
    assert eager['args'].iteration == 8
    worst = 0.0
    for part in ('generator', 'discriminator', 'embedder'):
        for k, v in eager[part].items():
            if v.dtype == torch.float32:
                d = ((graph[part][k].double() - v.double()).norm() / v.double().norm().clamp_min(1e-30)).item()
                worst = max(worst, d)
    print(f'[entry] --hip_graph vs eager after 8 iterations of run_epoch: worst state rel-L2 {worst:.2e}')
    assert worst < 1e-5, worst


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
