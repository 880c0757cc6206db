"""Network plugin registry -- the drop-in boundary (reference msmctts/networks/__init__.py:6-11).

``find_modules(conf)`` takes ``{name: {_name: ClassName, **kwargs}}`` (the ``task`` section of the
YAML), resolves each class by scanning this package exactly like the reference's ``module_search``
and returns ``[(name, instance)]``; keys starting with ``_`` are not passed to the constructor.
"""
import os

from ..utils.utils import module_search


def find_modules(conf):
    names, confs = zip(*conf.items())
    classes = module_search([c['_name'] for c in confs], os.path.dirname(__file__), __name__)
    return [(n, cls(**{k: v for k, v in c.items() if k[:1] != '_'})) for n, cls, c in zip(names, classes, confs)]
