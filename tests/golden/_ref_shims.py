"""Import shims that let the *unmodified* reference (/root/reference) run on CPU in the build
container.  Used ONLY by tests/golden/make_golden.py to produce fixtures; nothing on the GPU box
touches this file's code paths (the reference does not travel).

Why each shim exists:
  1. ``soundfile``      not installed; utils/utils.py:2,111 subclasses ``SoundFile``
  2. ``turtle.update``  stray import at vqgantts/msmc_vqgan.py:1 (needs tkinter)
  3. ``tensorboardX``   not installed; utils/logger.py:1
  4. ``librosa``        not installed; stft_loss.py:1,85 and utils/audio.py:1,8
  5. ``msmctts.networks.vqgantts.msmc_vqgan_speech``  missing file imported at msmc_vqgan_emb.py:11
  6. ``torch.stft``     reference calls it without ``return_complex`` (audio.py:399, stft_loss.py:21,95)
"""
import importlib.abc
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get('MSMC_REFERENCE_ROOT', '/root/reference')
_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


class _AliasFinder(importlib.abc.MetaPathFinder):
    """Resolve the missing ``msmc_vqgan_speech`` module to ``msmc_vqgan.py`` (same package)."""
    missing = 'msmctts.networks.vqgantts.msmc_vqgan_speech'

    def find_spec(self, fullname, path, target=None):
        if fullname != self.missing:
            return None
        src = os.path.join(REFERENCE_ROOT, 'msmctts', 'networks', 'vqgantts', 'msmc_vqgan.py')
        return importlib.util.spec_from_file_location(fullname, src)


def install():
    if getattr(install, 'done', False):
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError('reference tree not found at %s (fixtures can only be generated in the '
                           'build container)' % REFERENCE_ROOT)
    if _REPO not in sys.path:
        sys.path.insert(0, _REPO)
    from oracle.audio import slaney_mel_basis

    class _SoundFile(object):
        pass

    _stub('soundfile', SoundFile=_SoundFile)
    _stub('turtle', update=lambda *a, **k: None)

    class _Writer(object):
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

    _stub('tensorboardX', SummaryWriter=_Writer)

    def _mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, **kw):
        fmax = sr / 2.0 if fmax is None else fmax
        return slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax)

    def _pad_center(data, size, axis=-1, **kw):
        n = data.shape[axis]
        lpad = int((size - n) // 2)
        widths = [(0, 0)] * data.ndim
        widths[axis] = (lpad, int(size - n - lpad))
        return np.pad(data, widths, mode='constant')

    def _tiny(x):
        return np.finfo(np.asarray(x).dtype if np.issubdtype(np.asarray(x).dtype, np.floating)
                        else np.float32).tiny

    def _normalize(S, norm=np.inf, axis=0, **kw):
        mag = np.abs(S).astype(float)
        length = np.max(mag, axis=axis, keepdims=True) if norm == np.inf else \
            np.sum(mag ** norm, axis=axis, keepdims=True) ** (1.0 / norm)
        length[length < _tiny(S)] = 1.0
        return S / length

    filters = _stub('librosa.filters', mel=_mel)
    util = _stub('librosa.util', pad_center=_pad_center, tiny=_tiny, normalize=_normalize)
    _stub('librosa', filters=filters, util=util)

    sys.meta_path.insert(0, _AliasFinder())

    real_stft = torch.stft

    def stft_compat(input, n_fft, hop_length=None, win_length=None, window=None, center=True,
                    pad_mode='reflect', normalized=False, onesided=None, return_complex=None):
        if return_complex is None:
            out = real_stft(input, n_fft, hop_length, win_length, window, center, pad_mode,
                            normalized, onesided, return_complex=True)
            return torch.view_as_real(out)
        return real_stft(input, n_fft, hop_length, win_length, window, center, pad_mode,
                         normalized, onesided, return_complex=return_complex)

    torch.stft = stft_compat
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    install.done = True
