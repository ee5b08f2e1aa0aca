"""Feature matching over the discriminator's feature maps (reference API: criterions/featmat.py:4-29)."""
import torch.nn.functional as F
from torch import nn


class Wrapper:
    @staticmethod
    def get_args(parser):
        parser.add('--fm_weight', type=float, default=10.0)

    @staticmethod
    def get_net(args):
        return Criterion(args.fm_weight).to(args.device)


class Criterion(nn.Module):
    def __init__(self, fm_weight):
        super().__init__()
        self.fm_weight = fm_weight

    def forward(self, data_dict):
        fake, real = data_dict['fake_features'], data_dict['real_features']
        total = sum(F.l1_loss(f, r.detach()) for f, r in zip(fake, real))
        return {'feature_matching': total / len(fake) * self.fm_weight}
