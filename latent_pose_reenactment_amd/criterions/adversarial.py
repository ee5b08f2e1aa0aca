"""Hinge / relativistic GAN losses (reference API: criterions/adversarial.py:4-57)."""
import torch
from torch import nn


class Wrapper:
    @staticmethod
    def get_args(parser):
        parser.add('--gan_type', type=str, default='gan', help='gan|rgan|ragan')

    @staticmethod
    def get_net(args):
        return Criterion(args.gan_type).to(args.device)


class Criterion(nn.Module):
    def __init__(self, gan_type):
        super().__init__()
        if gan_type not in ('gan', 'rgan', 'ragan'):
            raise Exception('Incorrect `gan_type` argument')
        self.gan_type = gan_type

    def _preds(self, real, fake):
        if self.gan_type == 'gan':
            return real, fake
        if self.gan_type == 'rgan':
            return real - fake, fake - real
        return real - fake.mean(), fake - real.mean()

    def forward(self, data_dict):
        real = data_dict['real_score']
        real_pred, fake_pred_D = self._preds(real, data_dict['fake_score_D'])
        _, fake_pred_G = self._preds(real, data_dict['fake_score_G'])
        loss_D = torch.relu(1. - real_pred).mean() + torch.relu(1. + fake_pred_D).mean()
        if self.gan_type == 'gan':
            loss_G = -fake_pred_G.mean()
        else:
            loss_G = torch.relu(1. + real_pred).mean() + torch.relu(1. - fake_pred_G).mean()
        return {'adversarial_G': loss_G}, {'adversarial_D': loss_D}
