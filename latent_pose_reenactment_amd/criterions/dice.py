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
        if fake.is_cuda and fake.dtype == torch.float32 and fake.dim() == 4 and real.dim() == 4 and fake.shape[0] == real.shape[0] \
                and fake.shape[2:] == real.shape[2:] and fake.shape[1] in (1, real.shape[1]):
            return {'segmentation_dice': DiceFn.apply(fake, real.detach(), float(self.dice_weight))}      # lp_reduce_dice (+ _bwd): 2 + 1 launches
        raise RuntimeError('dice: expected fp32 CUDA tensors fake [B,1|C,H,W], real [B,C,H,W] (MI355X HIP path, no CPU fallback)')


class DiceFn(torch.autograd.Function):
    """-log(sum 2 f r / (sum f^2 + sum r^2)) * weight with the reference's broadcast (fake B x 1, real B x 3: numerator and sum r^2 over
    the expanded shape, sum f^2 over fake's own elements) -- criterions/dice.py:30-34 of the reference.  Gradient w.r.t. fake only."""

    @staticmethod
    def forward(ctx, fake, real, weight):
        from latent_pose_reenactment_amd import _lib
        f, r = fake.detach().contiguous(), real.contiguous().float()
        b, cf, h, w = f.shape
        buf = torch.empty(_lib.lib().lp_dice_partial_blocks() * 3 + 3, dtype=torch.float32, device=f.device)
        out, sums = buf[-3:-2], buf[-2:]
        _lib.check(_lib.lib().lp_reduce_dice(f.data_ptr(), r.data_ptr(), buf.data_ptr(), out.data_ptr(), sums.data_ptr(), b, cf, r.shape[1], h * w,
                                             weight, torch.cuda.current_stream().cuda_stream), 'lp_reduce_dice')
        ctx.save_for_backward(f, r, sums)
        ctx.weight = weight
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        from latent_pose_reenactment_amd import _lib
        f, r, sums = ctx.saved_tensors
        b, cf, h, w = f.shape
        df = torch.empty_like(f)
        g1 = g.reshape(1).contiguous().float()
        _lib.check(_lib.lib().lp_reduce_dice_bwd(f.data_ptr(), r.data_ptr(), sums.data_ptr(), g1.data_ptr(), df.data_ptr(), b, cf, r.shape[1], h * w,
                                                 ctx.weight, torch.cuda.current_stream().cuda_stream), 'lp_reduce_dice_bwd')
        return df, None, None
