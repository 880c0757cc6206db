"""Per-child optimizer bundle (drop-in for reference msmctts/trainers/optimizers/__init__.py:8-78).

One optimizer per top-level child of the task (``autoencoder``, ``discriminator``), configured by
``optimizer.<child>`` or ``optimizer._default``; ``zero_grad``/``step`` take child names.  On the GPU AdamW (and Adam
without weight decay, which is the same update) is ``HipAdamW``: gradient-norm clipping + update of every tensor of a
child in three launches (csrc/optim.hip), learning rate / step count on the device (hipGraph-replayable).
"""
import re

import torch
from torch.optim import Adam, AdamW

from .hip_adamw import HipAdamW
from .radam import RAdam



def get_optimizer(parameters, config, capturable=False):
    parameters = list(parameters)
    name = config._name
    args = (config.learning_rate, tuple(config.betas), config.eps, config.weight_decay)
    if name == 'RAdam':
        return RAdam(parameters, *args)
    cls = {'Adam': Adam, 'AdamW': AdamW}[name]
    fused = any(p.is_cuda for p in parameters)
    from ...hip import lib
    if (fused or lib._host_pointers_ok) and (name == 'AdamW' or config.weight_decay == 0):
        return HipAdamW(parameters, *args)
    if fused and capturable:         # hipGraph replay: step counters and lr live on the device
        dev = next(p.device for p in parameters if p.is_cuda)
        return cls(parameters, torch.tensor(float(config.learning_rate), device=dev), tuple(config.betas), config.eps,
                   config.weight_decay, fused=True, capturable=True)
    return cls(parameters, *args, fused=True) if fused else cls(parameters, *args)


def build_optimizer(model, config, capturable=False):
    optimizers, configs = {}, {}
    for child_name, child in model.named_children():
        if child_name in config:
            cfg = config[child_name]
        else:
            assert hasattr(config, '_default'), 'Both {} and _default not found'.format(child_name)
            cfg = config._default
        configs[child_name] = cfg
        params = child.parameters()
        if hasattr(cfg, 'parameters'):
            params = []
            for n, p in child.named_parameters():
                if re.match(cfg.parameters, n):
                    params.append(p)
                else:
                    p.requires_grad = False
        optimizers[child_name] = get_optimizer(params, cfg, capturable)
    return Optimizer(optimizers, configs)


class Optimizer(object):
    def __init__(self, optimizers_dict, config):
        self.optimizers, self.config = optimizers_dict, config

    def _names(self, names):
        if names is None:
            return tuple(self.optimizers)
        return names if isinstance(names, (list, tuple)) else [names]

    def state_dict(self):
        return {k: o.state_dict() for k, o in self.optimizers.items()}

    def load_state_dict(self, state):
        for k, o in self.optimizers.items():
            o.load_state_dict(state[k])

    def zero_grad(self, names=None):
        for k in self._names(names):
            self.optimizers[k].zero_grad()

    def step(self, names=None):
        for k in self._names(names):
            self.optimizers[k].step()

    def prepare(self, names=None, tag='graph'):
        """optimizers that keep device-side tables of their tensors build them now (called outside hipGraph capture) and keep
        them until ``release(names, tag)``"""
        for k in self._names(names):
            if hasattr(self.optimizers[k], 'prepare'):
                self.optimizers[k].prepare(tag)

    def release(self, names=None, tag='graph'):
        """the captured steps that asked for tables under ``tag`` were dropped"""
        for k in self._names(names):
            if hasattr(self.optimizers[k], 'release'):
                self.optimizers[k].release(tag)

    def clip_and_step(self, name, max_norm):
        """clip the global gradient norm of child ``name`` to ``max_norm`` and step it; returns the pre-clip norm
        (reference: clip_grad_norm_ then optimizer.step, msmctts_trainer.py:205-207)"""
        opt = self.optimizers[name]
        if isinstance(opt, HipAdamW):
            opt.step(max_norm=max_norm)
            return opt.grad_norm
        params = [p for g in opt.param_groups for p in g['params']]
        norm = torch.nn.utils.clip_grad_norm_(params, max_norm)
        opt.step()
        return norm
