"""Task container: the module whose children (``autoencoder``, ``discriminator``) key optimizers and
checkpoints (reference msmctts/tasks/__init__.py:9-43, base_task.py:6-33), built from a configuration or restored
from a checkpoint (``infer.py``)."""
from os.path import dirname

import torch

from ..utils.config import Config
from ..utils.utils import load_checkpoint, module_search
from .base_task import BaseTask


def load_task(checkpoint_path, config_path=None, mode='infer'):
    ckpt = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
    task = build_task(Config(config_path if config_path is not None else ckpt['config']), mode)
    load_checkpoint(ckpt, task)
    return task


def load_model(name, checkpoint_path, config_path=None):
    return getattr(load_task(checkpoint_path, config_path), name)


def build_task(config=None, mode='train', checkpoint=None, *args, **kwargs):
    assert config is not None or checkpoint is not None
    if checkpoint is not None:
        return load_task(checkpoint, config, mode)
    if isinstance(config, str):
        config = Config(config)
    assert type(config) == Config
    cls = module_search(config.task._name, dirname(__file__), __name__)
    return cls(config, mode=mode, *args, **kwargs)
