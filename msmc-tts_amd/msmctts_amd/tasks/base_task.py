"""Module container keyed by network name (reference msmctts/tasks/base_task.py:6-33)."""
import torch

from ..networks import find_modules


class BaseTask(torch.nn.Module):
    def __init__(self, config, mode='train'):
        super().__init__()
        self.config, self.mode = config, mode
        nets = config.task.network if hasattr(config.task, 'network') else \
            {k: v for k, v in config.task.items() if k[:1] != '_' and '_name' in v}
        for name, net in find_modules(nets):
            self.add_module(name, net)

    def forward(self, features):
        return {'train': self.train_step, 'infer': self.infer_step, 'debug': self.debug_step}[self.mode](features)

    def train_step(self, features):
        pass

    def infer_step(self, features):
        raise NotImplementedError

    def debug_step(self, features):
        pass
