"""Per-child optimizer bundle (drop-in for reference msmctts/trainers/optimizers/__init__.py:8-78).

One optimizer per top-level child of the task (``autoencoder``, ``discriminator``), configured by
``optimizer.<child>`` or ``optimizer._default``; ``zero_grad``/``step`` take child names.  On the GPU
AdamW runs as PyTorch's fused multi-tensor kernel (one launch per child instead of one per tensor).
"""
import re

import torch
from torch.optim import Adam, AdamW

from .radam import RAdam


def get_optimizer(parameters, config, capturable=False):
    parameters = list(parameters)
    name = config._name
    args = (config.learning_rate, tuple(config.betas), config.eps, config.weight_decay)
    if name == 'RAdam':
        return RAdam(parameters, *args)
    cls = {'Adam': Adam, 'AdamW': AdamW}[name]
    fused = any(p.is_cuda for p in parameters)
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
