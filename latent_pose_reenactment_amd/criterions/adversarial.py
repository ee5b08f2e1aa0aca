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
        fd, fg = data_dict['fake_score_D'], data_dict['fake_score_G']
        if self.gan_type == 'gan' and real.is_cuda and real.dtype == torch.float32 and real.shape == fd.shape == fg.shape:
            return {'adversarial_G': HingeGFn.apply(fg)}, {'adversarial_D': HingeDFn.apply(real, fd)}      # lp_reduce_hinge (+ _bwd)
        real_pred, fake_pred_D = self._preds(real, data_dict['fake_score_D'])
        _, fake_pred_G = self._preds(real, data_dict['fake_score_G'])
        loss_D = torch.relu(1. - real_pred).mean() + torch.relu(1. + fake_pred_D).mean()
        if self.gan_type == 'gan':
            loss_G = -fake_pred_G.mean()
        else:
            loss_G = torch.relu(1. + real_pred).mean() + torch.relu(1. - fake_pred_G).mean()
        return {'adversarial_G': loss_G}, {'adversarial_D': loss_D}


class HingeDFn(torch.autograd.Function):
    """loss_D = mean(relu(1 - real)) + mean(relu(1 + fake_D)) over the B critic scores (criterions/adversarial.py:41-44 of the reference,
    gan_type 'gan'): lp_reduce_hinge / lp_reduce_hinge_bwd.  loss_G and loss_D are SEPARATE autograd nodes on purpose: a joint node would
    make the whole generator/embedder graph reachable from loss_D (the engine walks edges even when a gradient is undefined, and every
    Python Function on the way would materialise zeros and run its backward)."""

    @staticmethod
    def forward(ctx, real, fake_d):
        from latent_pose_reenactment_amd import _lib
        r, fd = (t.detach().contiguous().reshape(-1) for t in (real, fake_d))
        out = torch.empty(2, dtype=torch.float32, device=r.device)
        _lib.check(_lib.lib().lp_reduce_hinge(r.data_ptr(), fd.data_ptr(), fd.data_ptr(), out.data_ptr(), r.numel(),
                                              torch.cuda.current_stream().cuda_stream), 'lp_reduce_hinge')
        ctx.save_for_backward(r, fd)
        ctx.shape = real.shape
        return out[1]

    @staticmethod
    def backward(ctx, gD):
        from latent_pose_reenactment_amd import _lib
        r, fd = ctx.saved_tensors
        n = r.numel()
        g = gD.reshape(1).contiguous().float()
        outs = [torch.empty(n, dtype=torch.float32, device=r.device) if nd else None for nd in ctx.needs_input_grad]
        p = lambda t: None if t is None else t.data_ptr()
        _lib.check(_lib.lib().lp_reduce_hinge_bwd(r.data_ptr(), fd.data_ptr(), None, g.data_ptr(), p(outs[0]), p(outs[1]), None, n,
                                                  torch.cuda.current_stream().cuda_stream), 'lp_reduce_hinge_bwd')
        return tuple(None if o is None else o.view(ctx.shape) for o in outs)


class HingeGFn(torch.autograd.Function):
    """loss_G = -mean(fake_G) (criterions/adversarial.py:46-47 of the reference)"""

    @staticmethod
    def forward(ctx, fake_g):
        from latent_pose_reenactment_amd import _lib
        fg = fake_g.detach().contiguous().reshape(-1)
        out = torch.empty(2, dtype=torch.float32, device=fg.device)
        _lib.check(_lib.lib().lp_reduce_hinge(fg.data_ptr(), fg.data_ptr(), fg.data_ptr(), out.data_ptr(), fg.numel(),
                                              torch.cuda.current_stream().cuda_stream), 'lp_reduce_hinge')
        ctx.save_for_backward(fg)
        ctx.shape = fake_g.shape
        return out[0]

    @staticmethod
    def backward(ctx, gG):
        from latent_pose_reenactment_amd import _lib
        (fg,) = ctx.saved_tensors
        n = fg.numel()
        g = gG.reshape(1).contiguous().float()
        d = torch.empty(n, dtype=torch.float32, device=fg.device)
        _lib.check(_lib.lib().lp_reduce_hinge_bwd(fg.data_ptr(), fg.data_ptr(), g.data_ptr(), None, None, None, d.data_ptr(), n,
                                                  torch.cuda.current_stream().cuda_stream), 'lp_reduce_hinge_bwd')
        return d.view(ctx.shape)
