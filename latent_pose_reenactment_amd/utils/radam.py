"""RAdam with the exact update rule of the reference's vendored optimizer (utils/radam.py:29-95,
degenerated_to_sgd=True): the finetuning config uses it with betas=(0, 0.999), eps=1e-5.  Same constructor and
state_dict layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter) so reference optimizer states load."""
import math

import torch
from torch.optim.optimizer import Optimizer


class RAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, degenerated_to_sgd=True):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError('invalid RAdam hyper-parameter')
        self.degenerated_to_sgd = degenerated_to_sgd
        # ``buffer``: the reference keeps a 10-entry (step, N_sma, step_size) cache in every param group (utils/radam.py:22-23,63-80)
        # and indexes it in step(); it is emitted here (unused: the rectification is recomputed) so that an optimizer state
        # written by this class loads into the reference's RAdam.
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      buffer=[[None, None, None] for _ in range(10)]))

    @staticmethod
    def rectification(step, beta1, beta2, degenerated_to_sgd=True):
        """-> (n_sma, step_size); step_size < 0 means 'skip the update'."""
        beta2_t = beta2 ** step
        n_max = 2.0 / (1.0 - beta2) - 1.0
        n_sma = n_max - 2.0 * step * beta2_t / (1.0 - beta2_t)
        if n_sma >= 5:
            step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2)) \
                / (1 - beta1 ** step)
        elif degenerated_to_sgd:
            step_size = 1.0 / (1 - beta1 ** step)
        else:
            step_size = -1.0
        return n_sma, step_size

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                g = p.grad.float()
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, dtype=torch.float32)
                    st['exp_avg_sq'] = torch.zeros_like(p, dtype=torch.float32)
                m, v = st['exp_avg'], st['exp_avg_sq']
                v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
                m.mul_(beta1).add_(g, alpha=1 - beta1)
                st['step'] += 1
                n_sma, step_size = self.rectification(st['step'], beta1, beta2, self.degenerated_to_sgd)
                if n_sma >= 5:
                    if group['weight_decay'] != 0:
                        p.add_(p, alpha=-group['weight_decay'] * group['lr'])
                    p.addcdiv_(m, v.sqrt().add_(group['eps']), value=-step_size * group['lr'])
                elif step_size > 0:
                    if group['weight_decay'] != 0:
                        p.add_(p, alpha=-group['weight_decay'] * group['lr'])
                    p.add_(m, alpha=-step_size * group['lr'])
        return loss
