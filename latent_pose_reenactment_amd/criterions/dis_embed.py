"""Embedder-vs-discriminator identity embedding matching (reference API: criterions/dis_embed.py:5-34)."""
import torch.nn.functional as F
from torch import nn


class Wrapper:
    @staticmethod
    def get_args(parser):
        parser.add('--dis_embed_weight', type=float, default=1e-2)

    @staticmethod
    def get_net(args):
        return Criterion(args.dis_embed_weight).to(args.device)


class Criterion(nn.Module):
    def __init__(self, dis_embed_weight):
        super().__init__()
        self.weight = dis_embed_weight

    def forward(self, data_dict):
        fake, real = data_dict['embeds_elemwise'], data_dict['real_embedding']
        if fake.dim() > 2:
            fake = fake[:, 0]
        if real.dim() > 2:
            real = real[:, 0]
        return {'embedding_matching': F.l1_loss(fake, real.detach()) * self.weight}
