"""pytest configuration: registers the ``gpu`` marker and puts the repo root and the product
package directory (``msmc-tts_amd/``, hyphenated on purpose, so not importable by name) on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'msmc-tts_amd'), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


import msmctts_amd  # noqa: E402,F401  -- before the first HIP call: sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (see its docstring)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _unbounded_tuning_budget():
    """the per-process budget of kernel-timing launches (hip/conv.py TUNE_BUDGET) protects long training runs; a test
    session touches more layer shapes than any of them and tests that force a kernel generation rely on the timing pass"""
    from msmctts_amd.hip import conv
    conv.TUNE_BUDGET[0] = 10 ** 9
    yield
