"""Trainer factory (reference msmctts/trainers/__init__.py:6-12)."""
from os.path import dirname

from ..utils.utils import module_search


def build_trainer(config, model, num_gpus=1, rank=0):
    kwargs = config.trainer.to_dict()
    cls = module_search(kwargs.pop('_name'), dirname(__file__), __name__)
    return cls(config, model, num_gpus=num_gpus, rank=rank, **kwargs)
