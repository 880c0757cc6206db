"""Rectified Adam (selectable as ``optimizer._name: RAdam``; reference trainers/optimizers/radam.py:8-85).

Written from the RAdam paper's update rule (Liu et al. 2020, Algorithm 2) in the variant the
reference uses: variance rectification when rho_t >= 5 (reference radam.py:63,73 -- 'more conservative' than the
paper's rho_t > 4; SGD-with-momentum step otherwise) and
weight decay applied as an L2 term to the parameter before the update.
"""
import math

import torch
from torch.optim.optimizer import Optimizer


class RAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.5, 0.9), eps=1e-8, weight_decay=0):      # (the reference class defaults, radam.py:11)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2 = group['betas']
            rho_inf = 2.0 / (1.0 - b2) - 1.0
            for p in group['params']:
                if p.grad is None:
                    continue
                g = p.grad.float()
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, dtype=torch.float32)
                    st['exp_avg_sq'] = torch.zeros_like(p, dtype=torch.float32)
                st['step'] += 1
                t = st['step']
                st['exp_avg_sq'].mul_(b2).addcmul_(g, g, value=1 - b2)
                st['exp_avg'].mul_(b1).add_(g, alpha=1 - b1)
                b2t = b2 ** t
                rho_t = rho_inf - 2.0 * t * b2t / (1.0 - b2t)
                w = p.float()
                if group['weight_decay'] != 0:
                    w.add_(w, alpha=-group['weight_decay'] * group['lr'])
                if rho_t >= 5:
                    rect = math.sqrt((1 - b2t) * (rho_t - 4) / (rho_inf - 4) * (rho_t - 2) / rho_t * rho_inf / (rho_inf - 2))
                    w.addcdiv_(st['exp_avg'], st['exp_avg_sq'].sqrt().add_(group['eps']),
                               value=-group['lr'] * rect / (1 - b1 ** t))
                else:
                    w.add_(st['exp_avg'], alpha=-group['lr'] / (1 - b1 ** t))
                p.copy_(w)
        return loss
