"""LR schedule factory (reference msmctts/trainers/lr_schedulers/__init__.py:6-9)."""
from .exponential_lr import ExponentialDecayLRScheduler

_REGISTRY = {'ExponentialDecayLRScheduler': ExponentialDecayLRScheduler}


def build_lr_scheduler(config):
    kwargs = config.to_dict() if hasattr(config, 'to_dict') else dict(config)
    return _REGISTRY[kwargs.pop('_name')](**kwargs)
