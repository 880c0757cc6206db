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
    """STFT magnitude front-end of the MRD discriminators (reference audio.py:379-419).

    The transform runs on the gfx950 kernels (msmctts_amd/hip/spectral.py): reflect-padded framing, the
    windowed DFT and the HTK-mel projection as fp32 GEMMs on the matrix cores, fused magnitude / log image.
    ``image_cl`` hands the discriminator its channels-last input directly; ``transform`` keeps the reference's
    (B, 2F, T') return layout.  The unused ``atan2`` phase of the reference (:405) is not computed.
    """

    def __init__(self, fft_size, hop_size, win_size, normalized=False, domain='linear', mel_scale=False,
                 sample_rate=24000, ref_level_db=20, min_level_db=-100):
        super().__init__()
        self.fft_size, self.hop_size, self.win_size = fft_size, hop_size, win_size
        self.ref_level_db, self.min_level_db = ref_level_db, min_level_db
        self.normalized, self.domain = normalized, domain
        assert (ref_level_db, min_level_db) == (20, -100), 'the image kernel fixes the reference levels'
        self.mel_scale = MelScale(n_mels=fft_size // 2 + 1, sample_rate=sample_rate,
                                  n_stft=fft_size // 2 + 1) if mel_scale else None
        self._consts = {}

    def consts(self, device):
        key = str(device)
        if key not in self._consts:
            from ..hip import spectral
            win = torch.hann_window(self.win_size)
            if self.win_size < self.fft_size:
                left = (self.fft_size - self.win_size) // 2
                win = torch.nn.functional.pad(win, (left, self.fft_size - self.win_size - left))
            dft = spectral.dft_basis(self.fft_size, win, self.normalized, device)
            fb = None
            if self.mel_scale is not None:
                F = self.fft_size // 2 + 1
                fb = spectral.projection(create_fb_matrix(F, self.mel_scale.f_min, self.mel_scale.f_max, F,
                                                          self.mel_scale.sample_rate), device)
            self._consts[key] = (dft, fb)
        return self._consts[key]

    def image_cl(self, x, dtype=torch.float32):
        """x (B, L) -> channels-last [B, F, T', 2] in ``dtype``: ch0 (mel-scaled) magnitude, ch1 normalised log-magnitude."""
        from ..hip import spectral
        dft, fb = self.consts(x.device)
        with torch.autocast(device_type=x.device.type, enabled=False):
            return spectral.mrd_image(x.float(), self.fft_size, self.hop_size, dft, fb, dtype)

    def front(self, x, dtype=torch.float32):
        """the same evaluation kept as an object (hip/spectral.py MrdFront): its image and intermediates can serve a later
        pass over some of the same waveform rows (``image_rows``)"""
        from ..hip import spectral
        dft, fb = self.consts(x.device)
        with torch.autocast(device_type=x.device.type, enabled=False):
            return spectral.mrd_front(x.float(), self.fft_size, self.hop_size, dft, fb, dtype)

    def transform(self, x):
        img = self.image_cl(x)                       # [B, F, T, 2]
        B, F, T, _ = img.shape
        if self.domain == 'linear':
            return img[..., 0], None
        if self.domain == 'log':
            return img[..., 1], None
        return img.permute(0, 3, 1, 2).reshape(B, 2 * F, T), None
