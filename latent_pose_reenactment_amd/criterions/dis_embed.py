"""Embedder-vs-discriminator identity embedding matching (reference API: criterions/dis_embed.py:5-34)."""
from torch import nn

from latent_pose_reenactment_amd.nn import hip_l1_mean


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
        return {'embedding_matching': hip_l1_mean(fake, real) * self.weight}          # lp_l1_fwd / lp_l1_bwd (round 5: no ATen l1_loss left)
