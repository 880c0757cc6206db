"""Spectral feature losses (drop-in for reference msmctts/trainers/criterions/stft_loss.py).

``MelLoss`` (:55-114) and ``MultiResolutionSTFTLoss`` (:117-173).  The reference rebuilds the librosa
mel basis and re-uploads it on every call because its cache test never hits (:84-87), and it
host-syncs on ``torch.min/max`` (:79-82); here the Slaney basis (same values: librosa.filters.mel
defaults, restated) and the window are built once per device and nothing syncs.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _hz_from_slaney_mel(m):
    f_sp, brk = 200.0 / 3, 1000.0
    brk_mel, step = brk / f_sp, np.log(6.4) / 27.0
    return np.where(m >= brk_mel, brk * np.exp(step * (m - brk_mel)), f_sp * m)


def _slaney_mel_from_hz(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, brk = 200.0 / 3, 1000.0
    brk_mel, step = brk / f_sp, np.log(6.4) / 27.0
    return np.where(f >= brk, brk_mel + np.log(np.maximum(f, 1e-10) / brk) / step, f / f_sp)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """float32 (n_mels, 1 + n_fft//2): Slaney-scale, area-normalised triangles (librosa default)."""
    bins = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    edges = _hz_from_slaney_mel(np.linspace(_slaney_mel_from_hz(fmin), _slaney_mel_from_hz(fmax), n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - bins[None, :]
    lower = -ramps[:-2] / width[:-1, None]
    upper = ramps[2:] / width[1:, None]
    fb = np.maximum(0.0, np.minimum(lower, upper))
    fb *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return fb.astype(np.float32)


class MelLoss(nn.Module):
    """L1 between log-mel spectrograms (reference stft_loss.py:55-114) on the gfx950 kernels: reflect-padded
    framing, windowed DFT and the Slaney mel projection as fp32 matrix-core GEMMs, fused magnitude and log."""

    def __init__(self, fft_size, hop_size, win_size, sample_rate, num_mels):
        super().__init__()
        self.sample_rate, self.fft_size, self.hop_size = sample_rate, fft_size, hop_size
        self.win_size, self.num_mels = win_size, num_mels
        self.fmin, self.fmax = 0, sample_rate // 2
        self._basis = torch.from_numpy(mel_filterbank(sample_rate, fft_size, num_mels, self.fmin, self.fmax))
        self._cache = {}

    def _consts(self, device):
        key = str(device)
        if key not in self._cache:
            from ...hip import spectral
            win = torch.hann_window(self.win_size)
            left = (self.fft_size - self.win_size) // 2
            win = F.pad(win, (left, self.fft_size - self.win_size - left))
            self._cache[key] = (spectral.dft_basis(self.fft_size, win, False, device),
                                spectral.projection(self._basis.t(), device))
        return self._cache[key]

    def mel_spectrogram(self, y, center=False):
        """y (B, L) -> (B, num_mels, T') log-mel."""
        from ...hip import spectral
        assert not center
        dft, mel = self._consts(y.device)
        split = getattr(self, 'hip_dtype', torch.float32) == torch.bfloat16 and spectral.SPLIT_BF16
        lm = spectral.log_mel(y.float(), self.fft_size, self.hop_size, dft, mel, self.num_mels, split=split)   # [B, 1, T', M]
        return lm.squeeze(1).transpose(1, 2)

    def forward(self, predicts, targets):
        from ...hip import losses as hiploss
        from ...hip import spectral
        with torch.autocast(device_type=predicts.device.type, enabled=False):
            if (spectral.FRONTS_LOCKSTEP and hiploss.usable(predicts, targets) and predicts.shape == targets.shape
                    and not targets.requires_grad):
                # both chains in lock step (hip/spectral.py log_mel_pair): five launches instead of ten
                dft, mel = self._consts(predicts.device)
                split = getattr(self, 'hip_dtype', torch.float32) == torch.bfloat16 and spectral.SPLIT_BF16
                a, b = spectral.log_mel_pair(predicts.float(), targets.float(), self.fft_size, self.hop_size, dft, mel,
                                             self.num_mels, split=split)
                a, b = a.squeeze(1).transpose(1, 2), b.squeeze(1).transpose(1, 2)
            else:
                a, b = self.mel_spectrogram(predicts), self.mel_spectrogram(targets)
            if hiploss.usable(a, b) and a.dtype == b.dtype == torch.float32:
                # mean |a - b| as the multi-tensor L1 kernel's one-member call (two launches, one backward) instead of the
                # operator chain's subtract / abs / mean and their four backward launches
                return hiploss.l1_sum([a], [b])
            return F.l1_loss(a, b)


class STFTLoss(nn.Module):
    """Spectral-convergence + log-magnitude terms of one resolution (reference stft_loss.py:117-150).  The
    magnitude comes from the HIP spectral front-end (framing + windowed-DFT GEMM on the fp32 matrix cores +
    magnitude kernel, hip/spectral.py) instead of ``torch.stft``."""

    def __init__(self, fft_size, hop_size, win_size, mel_scale=False, sample_rate=24000):
        super().__init__()
        assert not mel_scale, 'mel-scaled MR-STFT is not used by any shipped config'
        self.fft_size, self.hop_size, self.win_size = fft_size, hop_size, win_size
        self._cache = {}

    def _dft(self, device):
        key = str(device)
        if key not in self._cache:
            from ...hip import spectral
            win = torch.hann_window(self.win_size)
            left = (self.fft_size - self.win_size) // 2
            win = F.pad(win, (left, self.fft_size - self.win_size - left))     # torch.stft centres the window
            self._cache[key] = spectral.dft_basis(self.fft_size, win, False, device)
        return self._cache[key]

    def _mag(self, x):
        """x (B, L) -> (B, T', F) magnitudes of the centred (reflect-padded) STFT, clamp 1e-7 under the root."""
        from ...hip import spectral
        return spectral.stft_magnitude(x, self.fft_size, self.hop_size, self._dft(x.device), 1e-7)

    def forward(self, predicts, targets):
        with torch.autocast(device_type=predicts.device.type, enabled=False):
            p, t = self._mag(predicts), self._mag(targets)
            sc = torch.norm(t - p, p='fro') / torch.norm(t, p='fro')
            mag = F.l1_loss(torch.log(torch.clamp(p, min=1e-5, max=10)), torch.log(torch.clamp(t, min=1e-5, max=10)))
        return sc, mag


class MultiResolutionSTFTLoss(nn.Module):
    def __init__(self, fft_sizes=[1024, 2048, 512], win_sizes=[600, 1200, 300], hop_sizes=[120, 240, 60],
                 mel_scale=False, sample_rate=24000):
        super().__init__()
        self.loss_layers = nn.ModuleList([STFTLoss(n, h, w, mel_scale, sample_rate)
                                          for n, w, h in zip(fft_sizes, win_sizes, hop_sizes)])

    def forward(self, fake_signals, true_signals):
        sc, mg = zip(*[layer(fake_signals, true_signals) for layer in self.loss_layers])
        return {'sc_loss': sum(sc) / len(sc), 'mag_loss': sum(mg) / len(mg)}
