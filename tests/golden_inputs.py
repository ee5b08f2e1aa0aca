"""Seeded input batches shared by tests/golden/make_golden.py (reference side, build container) and the GPU tests (product side):
fixtures whose inputs are too large to store keep only their checksums."""
import torch


def metatrain_big_batch(size=128, b=16, k=4, seed=55):
    """the seeded batch of metatrain_step_128.npz (built identically by tests/test_metatrain_step.py): smooth per-frame content + 10 %
    noise -- white-noise frames make deep features nearly constant over the batch, i.e. train-mode BatchNorm ill-conditioned in ANY
    arithmetic -- and a blob mask"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)

    def frames(n):
        low = torch.rand(n, 3, 8, 8, generator=g)
        x = F.interpolate(low, size=(size, size), mode='bilinear', align_corners=False)
        return (x + 0.1 * torch.rand(n, 3, size, size, generator=g)).clamp(0, 1)
    mask = F.interpolate(torch.rand(b, 1, 4, 4, generator=g), size=(size, size), mode='bilinear', align_corners=False)
    data = {'enc_rgbs': frames(b * k).view(b, k, 3, size, size), 'pose_input_rgbs': frames(b).view(b, 1, 3, size, size),
            'target_rgbs': (frames(b) * mask).view(b, 1, 3, size, size)}
    target = {'real_segm': mask[:, None].expand(b, 1, 3, size, size).contiguous(), 'label': torch.randint(0, 5, (b,), generator=g)}
    return data, target
