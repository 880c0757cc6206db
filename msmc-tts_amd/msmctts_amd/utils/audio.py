"""Spectral front-ends on the hot path (drop-in for the used parts of reference msmctts/utils/audio.py).

``create_fb_matrix`` (:30-84), ``MelScale`` (:314-376) and ``TorchSTFT.transform`` (:379-419).  The
reference also computes an ``atan2`` phase nobody consumes (:405) and re-uploads the window on every
call (:400); here the window and filter bank are cached per device and the phase is not computed.
"""
import math

import torch
import torch.nn as nn


def create_fb_matrix(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None):
    """(n_freqs, n_mels) HTK-mel triangles clamped to [1e-6, 1]."""
    if norm is not None and norm != 'slaney':
        raise ValueError("norm must be one of None or 'slaney'")
    freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_lo = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_hi = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    f_pts = 700.0 * (10 ** (torch.linspace(m_lo, m_hi, n_mels + 2) / 2595.0) - 1.0)
    df = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - freqs.unsqueeze(1)
    fb = torch.clamp(torch.min((-1.0 * slopes[:, :-2]) / df[:-1], slopes[:, 2:] / df[1:]), 1e-6, 1)
    if norm == 'slaney':
        fb = fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)
    return fb


class MelScale(nn.Module):
    def __init__(self, n_mels=128, sample_rate=24000, f_min=0., f_max=None, n_stft=None):
        super().__init__()
        self.n_mels, self.sample_rate, self.f_min = n_mels, sample_rate, f_min
        self.f_max = f_max if f_max is not None else float(sample_rate // 2)
        assert f_min <= self.f_max
        self._fb = {}

    def filter_bank(self, n_freq, like):
        key = (n_freq, str(like.device), like.dtype)
        if key not in self._fb:
            self._fb[key] = create_fb_matrix(n_freq, self.f_min, self.f_max, self.n_mels, self.sample_rate).to(like)
        return self._fb[key]

    def forward(self, specgram):
        shape = specgram.size()
        s = specgram.reshape(-1, shape[-2], shape[-1])
        mel = torch.matmul(s.transpose(1, 2), self.filter_bank(s.size(1), s)).transpose(1, 2)
        return mel.reshape(shape[:-2] + mel.shape[-2:])


class TorchSTFT(nn.Module):
    def __init__(self, fft_size, hop_size, win_size, normalized=False, domain='linear', mel_scale=False,
                 sample_rate=24000, ref_level_db=20, min_level_db=-100):
        super().__init__()
        self.fft_size, self.hop_size, self.win_size = fft_size, hop_size, win_size
        self.ref_level_db, self.min_level_db = ref_level_db, min_level_db
        self.normalized, self.domain = normalized, domain
        self.mel_scale = MelScale(n_mels=fft_size // 2 + 1, sample_rate=sample_rate,
                                  n_stft=fft_size // 2 + 1) if mel_scale else None
        self._win = {}

    def window(self, like):
        key = (str(like.device), like.dtype)
        if key not in self._win:
            self._win[key] = torch.hann_window(self.win_size, dtype=like.dtype, device=like.device)
        return self._win[key]

    def transform(self, x):
        """x (B, L) -> (magnitude image, None); 'double' domain returns cat(mag, norm-log-mag) on dim 1."""
        with torch.autocast(device_type=x.device.type, enabled=False):      # spectra stay fp32 under bf16 autocast
            x = x.float()
            spec = torch.stft(x, self.fft_size, self.hop_size, self.win_size, self.window(x),
                              normalized=self.normalized, return_complex=True)
            mag = torch.sqrt(torch.clamp(spec.real ** 2 + spec.imag ** 2, min=1e-7))
            if self.mel_scale is not None:
                mag = self.mel_scale(mag)
            if self.domain == 'linear':
                return mag, None
            log_mag = 20 * torch.log10(mag) - self.ref_level_db
            log_mag = torch.clamp((log_mag - self.min_level_db) / -self.min_level_db, 0, 1)
            if self.domain == 'log':
                return log_mag, None
            return torch.cat((mag, log_mag), dim=1), None
