"""Oracle: spectral front-ends of the hot path (plain PyTorch fp32, CPU).

Restates
  * ``create_fb_matrix``           reference msmctts/utils/audio.py:30-84
  * ``MelScale.forward``           reference msmctts/utils/audio.py:348-376
  * ``TorchSTFT.transform``        reference msmctts/utils/audio.py:398-419 ('double' domain)
  * ``MelLoss.mel_spectrogram``    reference msmctts/trainers/criterions/stft_loss.py:76-108
  * ``stft`` / MR-STFT losses      reference msmctts/trainers/criterions/stft_loss.py:11-52,117-173
  * ``librosa.filters.mel``        third party (librosa>=0.8.0), restated from its published definition
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# filter banks
# ----------------------------------------------------------------------------
def htk_triangle_bank(n_freqs, n_mels, sample_rate, f_min=0.0, f_max=None):
    """(n_freqs, n_mels) HTK-mel triangles clamped to [1e-6, 1]  (audio.py:30-84).

    MRD calls it with n_mels == n_freqs (audio.py:393-396), i.e. an F x F matrix.
    """
    f_max = float(sample_rate // 2) if f_max is None else f_max
    bins = torch.linspace(0, sample_rate // 2, n_freqs)
    lo = 2595.0 * math.log10(1.0 + f_min / 700.0)
    hi = 2595.0 * math.log10(1.0 + f_max / 700.0)
    mel_pts = torch.linspace(lo, hi, n_mels + 2)
    hz_pts = 700.0 * (10 ** (mel_pts / 2595.0) - 1.0)
    width = hz_pts[1:] - hz_pts[:-1]
    gap = hz_pts.unsqueeze(0) - bins.unsqueeze(1)          # (n_freqs, n_mels+2)
    falling = (-1.0 * gap[:, :-2]) / width[:-1]
    rising = gap[:, 2:] / width[1:]
    return torch.clamp(torch.min(falling, rising), 1e-6, 1)


def _slaney_hz(mels):
    f_sp = 200.0 / 3
    hz = f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(mels >= min_log_mel,
                    min_log_hz * np.exp(logstep * (mels - min_log_mel)), hz)


def _slaney_mel(hz):
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    hz = np.asarray(hz, dtype=np.float64)
    return np.where(hz >= min_log_hz,
                    min_log_mel + np.log(np.maximum(hz, 1e-10) / min_log_hz) / logstep,
                    hz / f_sp)


def slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax):
    """(n_mels, 1 + n_fft//2) float32: librosa.filters.mel(htk=False, norm='slaney').

    Third-party arithmetic (librosa>=0.8.0), called positionally at stft_loss.py:85.
    """
    n_bins = 1 + n_fft // 2
    fft_hz = np.linspace(0.0, float(sr) / 2, n_bins)
    mel_edges = np.linspace(_slaney_mel(fmin), _slaney_mel(fmax), n_mels + 2)
    hz_edges = _slaney_hz(mel_edges)
    step = np.diff(hz_edges)
    ramps = hz_edges[:, None] - fft_hz[None, :]
    w = np.zeros((n_mels, n_bins), dtype=np.float64)
    for m in range(n_mels):
        lower = -ramps[m] / step[m]
        upper = ramps[m + 2] / step[m + 1]
        w[m] = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (hz_edges[2:n_mels + 2] - hz_edges[:n_mels]))[:, None]
    return w.astype(np.float32)


# ----------------------------------------------------------------------------
# MRD front-end  (audio.py:398-419, 348-376)
# ----------------------------------------------------------------------------
_FB_CACHE = {}


def mrd_spectrogram(wav, hop, sample_rate=24000, mel_scale=True):
    """wav (B, L) -> (B, 2, F, T') with channel 0 = magnitude, 1 = normalised log-magnitude.

    n_fft = win = 4*hop, hann, centred reflect pad, ``normalized=True`` (discriminator.py:86-90).
    """
    n_fft = 4 * hop
    win = torch.hann_window(n_fft, dtype=wav.dtype, device=wav.device)
    spec = torch.stft(wav, n_fft, hop, n_fft, win, normalized=True, return_complex=True)
    mag = torch.sqrt(torch.clamp(spec.real ** 2 + spec.imag ** 2, min=1e-7))   # (B, F, T')
    if mel_scale:
        n_freq = mag.shape[1]
        key = (n_freq, sample_rate)
        if key not in _FB_CACHE:
            _FB_CACHE[key] = htk_triangle_bank(n_freq, n_freq, sample_rate)
        fb = _FB_CACHE[key].to(mag)
        mag = torch.matmul(mag.transpose(1, 2), fb).transpose(1, 2)
    log_mag = 20 * torch.log10(mag) - 20
    log_mag = torch.clamp((log_mag - (-100)) / 100, 0, 1)
    return torch.stack((mag, log_mag), dim=1)


# ----------------------------------------------------------------------------
# MelLoss  (stft_loss.py:55-114)
# ----------------------------------------------------------------------------
_MEL_CACHE = {}


def mel_loss_spectrogram(y, fft_size, hop_size, win_size, sample_rate, num_mels):
    """y (B, L) -> (B, num_mels, T') log-mel (stft_loss.py:76-108)."""
    key = (sample_rate, fft_size, num_mels)
    if key not in _MEL_CACHE:
        _MEL_CACHE[key] = torch.from_numpy(
            slaney_mel_basis(sample_rate, fft_size, num_mels, 0, sample_rate // 2))
    basis = _MEL_CACHE[key].to(y)
    pad = int((fft_size - hop_size) / 2)
    y = F.pad(y.unsqueeze(1), (pad, pad), mode='reflect').squeeze(1)
    win = torch.hann_window(win_size, dtype=y.dtype, device=y.device)
    spec = torch.stft(y, fft_size, hop_length=hop_size, win_length=win_size, window=win,
                      center=False, normalized=False, onesided=True, return_complex=True)
    spec = torch.sqrt(spec.real ** 2 + spec.imag ** 2 + 1e-9)
    spec = torch.matmul(basis, spec)
    return torch.log(torch.clamp(spec, min=1e-5))


def mel_loss(predict, target, sample_rate=24000, fft_size=None, hop_size=None, win_size=None,
             num_mels=128):
    """L1 between log-mels; defaults follow msmctts_trainer.py:101-110."""
    win_size = sample_rate // 20 if win_size is None else win_size
    hop_size = sample_rate // 80 if hop_size is None else hop_size
    fft_size = (2048 if win_size > 1024 else 1024) if fft_size is None else fft_size
    a = mel_loss_spectrogram(predict, fft_size, hop_size, win_size, sample_rate, num_mels)
    b = mel_loss_spectrogram(target, fft_size, hop_size, win_size, sample_rate, num_mels)
    return F.l1_loss(a, b)


# ----------------------------------------------------------------------------
# MR-STFT loss  (stft_loss.py:11-52, 117-173)
# ----------------------------------------------------------------------------
def _stft_mag(x, fft_size, hop_size, win_size):
    win = torch.hann_window(win_size, dtype=x.dtype, device=x.device)
    s = torch.stft(x, fft_size, hop_size, win_size, win, return_complex=True)
    return torch.sqrt(torch.clamp(s.real ** 2 + s.imag ** 2, min=1e-7)).transpose(2, 1)


def mr_stft_loss(fake, true, fft_sizes=(1024, 2048, 512), win_sizes=(600, 1200, 300),
                 hop_sizes=(120, 240, 60)):
    """-> {'sc_loss', 'mag_loss'} averaged over resolutions (stft_loss.py:152-172)."""
    sc, mg = [], []
    for n, w, h in zip(fft_sizes, win_sizes, hop_sizes):
        p = _stft_mag(fake, n, h, w)
        t = _stft_mag(true, n, h, w)
        sc.append(torch.norm(t - p, p='fro') / torch.norm(t, p='fro'))
        mg.append(F.l1_loss(torch.log(torch.clamp(p, min=1e-5, max=10)),
                            torch.log(torch.clamp(t, min=1e-5, max=10))))
    return {'sc_loss': sum(sc) / len(sc), 'mag_loss': sum(mg) / len(mg)}
