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
            loss_G, loss_D = HingeFn.apply(real, fd, fg)          # lp_reduce_hinge: one launch for both losses (+ one for the backward)
            return {'adversarial_G': loss_G}, {'adversarial_D': loss_D}
        real_pred, fake_pred_D = self._preds(real, data_dict['fake_score_D'])
        _, fake_pred_G = self._preds(real, data_dict['fake_score_G'])
        loss_D = torch.relu(1. - real_pred).mean() + torch.relu(1. + fake_pred_D).mean()
        if self.gan_type == 'gan':
            loss_G = -fake_pred_G.mean()
        else:
            loss_G = torch.relu(1. + real_pred).mean() + torch.relu(1. - fake_pred_G).mean()
        return {'adversarial_G': loss_G}, {'adversarial_D': loss_D}


class HingeFn(torch.autograd.Function):
    """(loss_G, loss_D) = (-mean(fake_G), mean(relu(1 - real)) + mean(relu(1 + fake_D))) over the B critic scores --
    criterions/adversarial.py:41-52 of the reference, gan_type 'gan'."""

    @staticmethod
    def forward(ctx, real, fake_d, fake_g):
        from latent_pose_reenactment_amd import _lib
        r, fd, fg = (t.detach().contiguous().reshape(-1) for t in (real, fake_d, fake_g))
        out = torch.empty(2, dtype=torch.float32, device=r.device)
        _lib.check(_lib.lib().lp_reduce_hinge(r.data_ptr(), fd.data_ptr(), fg.data_ptr(), out.data_ptr(), r.numel(),
                                              torch.cuda.current_stream().cuda_stream), 'lp_reduce_hinge')
        ctx.save_for_backward(r, fd)
        ctx.shape = real.shape
        # the two losses are backpropagated separately (loss_G.backward, then loss_D.backward: holycow.py:239-250): an output whose
        # gradient is absent must arrive as None -- a materialised zero would be propagated through the whole generator/embedder graph
        ctx.set_materialize_grads(False)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, gG, gD):
        from latent_pose_reenactment_amd import _lib
        r, fd = ctx.saved_tensors
        n = r.numel()
        # input i is reached only through the loss whose gradient is present: real, fake_d <- loss_D; fake_g <- loss_G
        need = [ctx.needs_input_grad[0] and gD is not None, ctx.needs_input_grad[1] and gD is not None, ctx.needs_input_grad[2] and gG is not None]
        if not any(need):
            return None, None, None
        g1 = None if gG is None else gG.reshape(1).contiguous().float()
        g2 = None if gD is None else gD.reshape(1).contiguous().float()
        outs = [torch.empty(n, dtype=torch.float32, device=r.device) if nd else None for nd in need]
        p = lambda t: None if t is None else t.data_ptr()
        _lib.check(_lib.lib().lp_reduce_hinge_bwd(r.data_ptr(), fd.data_ptr(), p(g1), p(g2), p(outs[0]), p(outs[1]), p(outs[2]), n,
                                                  torch.cuda.current_stream().cuda_stream), 'lp_reduce_hinge_bwd')
        return tuple(None if o is None else o.view(ctx.shape) for o in outs)
