"""Segmentation dice loss (reference API: criterions/dice.py:4-39) including its 1-channel vs 3-channel broadcast:
numerator and sum(real^2) run over the B x 3 expanded mask, sum(fake^2) over B x 1."""
import torch
from torch import nn


class Wrapper:
    @staticmethod
    def get_args(parser):
        parser.add('--dice_weight', type=float, default=1)

    @staticmethod
    def get_net(args):
        return Criterion(args.dice_weight).to(args.device)


class Criterion(nn.Module):
    def __init__(self, dice_weight):
        super().__init__()
        self.dice_weight = dice_weight

    def forward(self, data_dict):
        fake, real = data_dict['fake_segm'], data_dict['real_segm']
        if fake.dim() > 4:
            fake = fake[:, 0]
        if real.dim() > 4:
            real = real[:, 0]
        overlap = (2 * fake * real).sum()
        energy = (fake ** 2).sum() + (real ** 2).sum()
        return {'segmentation_dice': -torch.log(overlap / energy) * self.dice_weight}
