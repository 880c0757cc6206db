"""YAML configuration surface (drop-in for reference msmctts/utils/config.py:6-110).

Same observable behaviour: attribute access on nested dicts, the string 'none' (any case) becomes
None, values are merged recursively over the defaults, and scientific notation such as ``2e-4``
parses as a float (PyYAML's own resolver would give a string).
"""
import os
import re

import yaml

DEFAULTS = {
    'id': 'null',
    'save_checkpoint_dir': '',
    'pretrain_checkpoint_path': '',
    'restore_checkpoint_path': '',
    'resume_training': True,
    'training_steps': 1000000,
    'iters_per_checkpoint': 50000,
    'seed': 1234,
    'cudnn': {'enabled': True, 'benchmark': False},
    'distributed': {'dist_backend': 'nccl', 'dist_url': 'tcp://localhost:54321'},
}

_FLOAT = re.compile(r'''^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                       |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                       |\.[0-9_]+(?:[eE][-+][0-9]+)?
                       |[-+]?\.(?:inf|Inf|INF)
                       |\.(?:nan|NaN|NAN))$''', re.X)


class _Loader(yaml.SafeLoader):
    pass


_Loader.add_implicit_resolver('tag:yaml.org,2002:float', _FLOAT, list('-+0123456789.'))


def read_yaml(path):
    with open(path) as f:
        return yaml.load(f, Loader=_Loader)


def _wrap(value):
    if isinstance(value, ConfigItem):
        return ConfigItem(value.to_dict())
    if isinstance(value, dict):
        return ConfigItem(value)
    if isinstance(value, (list, tuple)):
        return [ConfigItem(v) if isinstance(v, dict) else v for v in value]
    if isinstance(value, str) and value.lower() == 'none':
        return None
    return value


class ConfigItem(dict):
    """dict with attribute access; nested dicts are wrapped on construction."""
    __slots__ = ()

    def __init__(self, mapping=None):
        super().__init__()
        if isinstance(mapping, ConfigItem):
            mapping = mapping.to_dict()
        for key, value in (mapping or {}).items():
            self[key] = _wrap(value)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def to_dict(self, recursive=True):
        return {k: (v.to_dict(True) if recursive and isinstance(v, ConfigItem) else v) for k, v in self.items()}

    def update(self, other):
        for key, value in other.items():
            if key in self and isinstance(value, dict) and isinstance(self[key], ConfigItem):
                self[key].update(value)
            else:
                self[key] = value


class Config(ConfigItem):
    """Config(path_or_dict): defaults overlaid with the YAML file / dict."""

    def __init__(self, source):
        super().__init__(DEFAULTS)
        if isinstance(source, str):
            if not os.path.isfile(source):
                raise AssertionError('config file not found: %s' % source)
            source = read_yaml(source)
        if isinstance(source, dict) and not isinstance(source, ConfigItem):
            source = ConfigItem(source)
        if not isinstance(source, ConfigItem):
            raise AssertionError('Config takes a YAML path or a dict')
        self.update(source)
