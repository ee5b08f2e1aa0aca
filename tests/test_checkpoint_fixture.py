"""The on-disk checkpoint format (reference utils/utils.py:251-398, drive.py:48-88) pinned by a file the REFERENCE's own
save_model wrote (tests/golden/reference_checkpoint_small.pth, produced by make_golden.py::make_checkpoint; it pickles tensors,
dicts and an argparse.Namespace only).  CPU: load_model_from_checkpoint restores every weight, both optimizer states and the
fine-tuning structure change.  GPU: drive.load_for_inference + one drive_frame on it, checked against the oracle."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'latent_pose_reenactment_amd')
for p in (PKG, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
CKPT = os.path.join(ROOT, 'tests', 'golden', 'reference_checkpoint_small.pth')


def _register():
    import tiny_embedder
    tiny_embedder.register()


def test_reference_written_checkpoint_loads(monkeypatch):
    _register()
    from utils import utils
    ck = utils.torch_load(CKPT)
    assert set(ck) == {'embedder', 'generator', 'discriminator', 'optimizer_G', 'optimizer_D', 'running_averages', 'args'}
    args = copy.copy(ck['args'])
    args.device = 'cpu'
    E, G, D, ra, saved_args, oG, oD = utils.load_model_from_checkpoint(ck, args)
    for name, mod in (('embedder', E), ('generator', G), ('discriminator', D)):
        sd = mod.state_dict()
        assert list(sd) == list(ck[name]), name                       # same keys in the same order
        for k, v in ck[name].items():
            assert torch.equal(sd[k], v), (name, k)
    assert set(ra) == {'embedder', 'generator'} and list(ra['generator']) == list(ck['generator'])
    # optimizer state in the reference's layout (positional params, per-param step / exp_avg / exp_avg_sq) is restored
    for opt, key in ((oG, 'optimizer_G'), (oD, 'optimizer_D')):
        st = opt.state_dict()
        assert len(st['state']) == len(ck[key]['state'])
        for i, s in ck[key]['state'].items():
            assert torch.equal(st['state'][i]['exp_avg'], s['exp_avg']) and float(st['state'][i]['step']) == float(s['step'])
    assert saved_args.iteration == 1234
    # the file is a FINE-TUNED checkpoint (what drive.py consumes): the structures follow (noBottleneck.py:139-163, no_landmarks.py:110-136)
    assert 'identity_embedding' in G.state_dict() and D.embed.weight_orig.shape[0] == 1 and E.finetuning and G.finetuning


@pytest.mark.gpu
def test_drive_frame_on_reference_written_checkpoint(monkeypatch):
    monkeypatch.setenv('LP_PREC', 'bf16x3')
    _register()
    import drive                      # (the entry script disables autograd globally at import, like the reference's drive.py:14)
    torch.set_grad_enabled(True)
    from oracle import lp_oracle as O
    with torch.no_grad():
        E, G, saved_args = drive.load_for_inference(CKPT, '/nonexistent', 'cuda:0')
        assert not G.training and G.finetuning
        ck = torch.load(CKPT, map_location='cpu', weights_only=False)
        g = torch.Generator().manual_seed(3)
        frame = torch.rand(1, 1, 3, 32, 32, generator=g)
        out = drive.drive_frame(E, G, {'pose_input_rgbs': frame.cuda()})
        assert out.shape == (32, 64, 3) and out.dtype == torch.uint8
        # oracle: EMA generator weights of the checkpoint, identity embedding as enable_finetuning() left it, eval mode
        sd = {k: v.clone() for k, v in G.state_dict().items()}
        sd = {k: v.cpu() for k, v in sd.items()}
        for k, v in ck['running_averages']['generator'].items():
            assert torch.equal(sd[k], v), k
        pose = E.cpu().pose_encoder(frame[:, 0].mean(dim=(2, 3)))
        rgb, _ = O.generator_forward(sd, sd['identity_embedding'], pose, num_channels=4, max_num_channels=16, image_size=32, train=False)
        want = rgb[0].permute(1, 2, 0).clamp(0, 1).mul(255).byte()
        diff = (out[:, 32:].cpu().int() - want.int()).abs()
        assert diff.max().item() <= 1 and (diff > 0).float().mean().item() < 0.02, (diff.max().item(), (diff > 0).float().mean().item())


@pytest.mark.gpu
def test_drive_batch_of_frames_equals_single_frames(monkeypatch):
    """drive loop with B > 1 driving frames per generator call: same frames as B single-frame calls, results stay on the device"""
    monkeypatch.setenv('LP_PREC', 'bf16x3')
    _register()
    import drive
    torch.set_grad_enabled(True)
    with torch.no_grad():
        E, G, _ = drive.load_for_inference(CKPT, '/nonexistent', 'cuda:0')
        g = torch.Generator().manual_seed(4)
        frames = torch.rand(3, 1, 3, 32, 32, generator=g).cuda()
        batch = drive.drive_frames(E, G, {'pose_input_rgbs': frames})
        assert batch.is_cuda and batch.shape == (3, 32, 64, 3) and batch.dtype == torch.uint8
        for i in range(3):
            one = drive.drive_frame(E, G, {'pose_input_rgbs': frames[i:i + 1]})
            assert (batch[i].int() - one.int()).abs().max().item() <= 1
