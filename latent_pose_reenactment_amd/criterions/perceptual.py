"""VGG-19 perceptual criterion (reference API: criterions/perceptual.py:4-33)."""
from torch import nn

from .common.perceptual_loss import PerceptualLoss


class Wrapper:
    @staticmethod
    def get_args(parser):
        parser.add('--perc_weight', type=float, default=1e-2)

    @staticmethod
    def get_net(args):
        return Criterion(args.perc_weight, args.vgg_weights_dir, getattr(args, 'synthetic_vgg_seed', None)).to(args.device)


class Criterion(nn.Module):
    independent_branch = True          # reads fake / target images only: may run beside the discriminator pass (streams.py)

    def __init__(self, perc_weight, vgg_weights_dir, synthetic_seed=None):
        super().__init__()
        self.perceptual_crit = PerceptualLoss(perc_weight, vgg_weights_dir, 'caffe', synthetic_seed).eval()

    def precompute_targets(self, data_dict):
        """features of the target image, ahead of the generator (same stream as the later ``forward`` call: streams.py)"""
        real = data_dict['target_rgbs']
        real = real[:, 0] if real.dim() > 4 else real
        self.__dict__['_taps_t'] = (real.data_ptr(), real._version, self.perceptual_crit.target_features(real))

    def forward(self, data_dict):
        fake, real = data_dict['fake_rgbs'], data_dict['target_rgbs']
        if fake.dim() > 4:
            fake = fake[:, 0]
        if real.dim() > 4:
            real = real[:, 0]
        pre = self.__dict__.pop('_taps_t', None)
        taps_t = pre[2] if pre is not None and pre[:2] == (real.data_ptr(), real._version) else None
        return {'VGG': self.perceptual_crit(fake, real, taps_t)}
