"""CPU: the PRODUCT Python path over the kernel *interpreter* build (tests/emu) -- host logic,
autograd wiring and kernel indexing/tiling logic against the reference fixtures.  The real
gfx950 library is exercised by tests/test_gpu_parity.py (-m gpu)."""
import os
import subprocess

import pytest
import torch

import _parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, 'tests', 'emu', 'libmsmc_emu.so')


@pytest.fixture(scope='module', autouse=True)
def emulator():
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'tests', 'emu')])
    from msmctts_amd.hip import lib
    lib.use_library_for_tests(EMU)
    assert lib.backend() == 'emu'
    torch.set_num_threads(4)
    yield


def test_vq_fixture_cases():
    _parity.check_vq_cases('cpu')


def test_modules_match_reference():
    _parity.check_modules('cpu')


def test_train_steps_match_reference():
    _parity.check_train_steps('cpu')
