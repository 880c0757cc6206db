"""VQ-GAN training step (re-expression of reference msmctts/trainers/msmctts_trainer.py:39-219).

Same phase logic, loss weights, loss-dictionary keys and optimizer order as the reference:
warm-up (no vocoder) for ``iteration < warmup_steps``; at ``iteration == warmup_steps`` the vocoder
runs but no GAN/STFT loss is taken (strict ``>``, :146); afterwards mel/STFT loss, D step (fake
detached + real, LSGAN), then the G step against the *updated* D (adversarial + feature matching),
gradient clipping on the autoencoder only.

Differences that change no result:
* in the G step the reference also back-propagates into D's weights and throws those gradients
  away; here D is frozen for that pass, D(real) runs without a graph and nothing of D is
  all-reduced on the G backward;
* loss values stay on the device (0-dim tensors, ``float(v)`` to read) instead of ~15 host syncs.
"""
import contextlib
import random

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..utils.utils import get_mask_from_lengths
from .base_trainer import BaseTrainer
from .criterions.stft_loss import MelLoss, MultiResolutionSTFTLoss


class QuantizerLoss(nn.Module):
    def __init__(self, lambda_vq=1, lambda_pr=1):
        super().__init__()
        self.lambda_vq, self.lambda_pr = lambda_vq, lambda_pr

    def forward(self, outputs):
        loss = {'vq_loss': 0}
        diffs = outputs['encoder_diffs']
        if not isinstance(diffs, (tuple, list)):
            diffs = [diffs]
        for i, terms in enumerate(diffs):
            length = outputs['encoder_lengths'][i]
            terms = terms if isinstance(terms, (tuple, list)) else [terms]
            for j, term in enumerate(terms):
                pad = get_mask_from_lengths(length.to(term.device), term.shape[1]).unsqueeze(-1)
                term = term.masked_fill(pad, 0).sum() / length.sum() / term.shape[2]
                loss['latent_loss_{}_{}'.format(i, j)] = term
                loss['vq_loss'] = loss['vq_loss'] + self.lambda_vq * term
        dd = outputs.get('decoder_diffs')
        if isinstance(dd, dict):
            dd = dict(dd)
            loss['vq_loss'] = loss['vq_loss'] + self.lambda_pr * dd.pop('total_loss')
            loss.update(dd)
        return loss


@contextlib.contextmanager
def _frozen(module):
    flags = [(p, p.requires_grad) for p in module.parameters()]
    for p, _ in flags:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p, f in flags:
            p.requires_grad_(f)


class VQGANTrainer(BaseTrainer):
    def __init__(self, config, model, num_gpus=1, rank=0, warmup_steps=0, lambda_frame=1.0,
                 eval_inteval_iters=1000, grad_clip_thresh=1.0, sample_lengths=24000, lambda_vq=1, lambda_pr=1,
                 lambda_fm=2, lambda_stft=45, stft_loss_func='mel_loss', stft_loss_config=None):
        super().__init__(config, model, num_gpus, rank)
        self.lambda_frame, self.warmup_steps = lambda_frame, warmup_steps
        self.frameshift = self.config.dataset.frameshift[self.config.dataset.feature.index('mel')]
        self.frame_lengths = -1 if sample_lengths == -1 else sample_lengths // self.frameshift
        self.eval_inteval_iters, self.grad_clip_thresh = eval_inteval_iters, grad_clip_thresh
        self.vq_criterion = QuantizerLoss(lambda_vq=lambda_vq, lambda_pr=lambda_pr)
        self.sample_lengths, self.lambda_fm, self.lambda_stft = sample_lengths, lambda_fm, lambda_stft
        if stft_loss_func == 'mel_loss':
            sr = config.dataset.samplerate
            kw = dict(sample_rate=sr, win_size=sr // 20, hop_size=sr // 80, num_mels=128)
            kw['fft_size'] = 2048 if kw['win_size'] > 1024 else 1024
            if stft_loss_config is not None:
                kw.update(stft_loss_config)
            self.stft_criterion = MelLoss(**kw)
        elif stft_loss_func == 'mr_stft':
            self.stft_criterion = MultiResolutionSTFTLoss(**dict(stft_loss_config or {}))
        self.rng = random              # python global RNG, like the reference (:214); tests inject their own
        self._amp_applied = None
        self.amp_dtype = None          # e.g. torch.bfloat16: autocast for the GEMM/conv bodies (VQ search stays fp32)

    def random_select(self, mel_length):
        lengths = mel_length.tolist() if torch.is_tensor(mel_length) else list(mel_length)
        frame_windows, sample_windows = [], []
        for n in lengths:
            start = self.rng.randrange(max(1, int(n) - self.frame_lengths))
            end = start + self.frame_lengths
            frame_windows.append((start, end))
            sample_windows.append((start * self.frameshift, end * self.frameshift))
        return frame_windows, sample_windows

    def _amp(self):
        if self._amp_applied is not self.amp_dtype:          # tell the HIP conv stacks their compute dtype
            for m in self.model.modules():
                if hasattr(m, 'hip_dtype'):
                    m.hip_dtype = self.amp_dtype or torch.float32
            self._amp_applied = self.amp_dtype
        if self.amp_dtype is None:
            return contextlib.nullcontext()
        device_type = next(self.model.parameters()).device.type
        return torch.autocast(device_type=device_type, dtype=self.amp_dtype)

    def train_step(self, batch, iteration):
        losses = {}
        mel, mel_length, wav = batch['mel'], batch['mel_length'], batch['wav']
        ae, disc = self.model.autoencoder, getattr(self.model, 'discriminator', None)
        if iteration < self.warmup_steps:
            with self._amp():
                out = ae(mel, mel_length, warmup=True)
        else:
            frame_windows, sample_windows = self.random_select(batch.get('mel_length_host', mel_length))
            target = torch.stack([wav[i, s:e] for i, (s, e) in enumerate(sample_windows)], dim=0)
            with self._amp():
                out = ae(mel, mel_length, warmup=False, window=frame_windows)

        vq = self.vq_criterion(out)
        losses.update(vq)
        g_loss = vq['vq_loss']

        if 'mel_outputs' in out:
            ml = F.mse_loss(mel, out['mel_outputs'].float(), reduction='none')
            ml = ml.masked_fill(get_mask_from_lengths(mel_length, ml.shape[1]).unsqueeze(-1), 0)
            ml = ml.sum() / mel_length.sum() / ml.shape[2]
            losses['frame_loss'] = ml
            g_loss = g_loss + self.lambda_frame * ml

        if iteration > self.warmup_steps:
            predict = out['decoder_outputs'].squeeze(-1).float()
            target = target.squeeze(-1)
            st = self.stft_criterion(predict, target)
            if isinstance(st, dict):
                for name, term in st.items():
                    losses[name] = term
                st = sum(st.values())
            losses['stft_loss'] = st
            g_loss = g_loss + self.lambda_stft * st

            # ---- discriminator step
            with self._amp():
                fake_scores, _ = disc(predict.detach())
                real_scores, _ = disc(target)
            d_real = sum(F.mse_loss(r.float(), torch.ones_like(r, dtype=torch.float32)) for r in real_scores)
            d_fake = sum(F.mse_loss(f.float(), torch.zeros_like(f, dtype=torch.float32)) for f in fake_scores)
            d_loss = d_real + d_fake
            losses['d_loss_real'], losses['d_loss_fake'], losses['d_loss'] = d_real, d_fake, d_loss
            self.optimizer.zero_grad(['discriminator'])
            d_loss.backward()
            self._sync_grads()
            self.optimizer.step(['discriminator'])

            # ---- generator step (against the updated D; D's own gradients are not needed)
            with _frozen(disc), self._amp():
                fake_scores, fake_feats = disc(predict)
                with torch.no_grad():
                    _, real_feats = disc(target)
            adv = sum(F.mse_loss(f.float(), torch.ones_like(f, dtype=torch.float32)) for f in fake_scores)
            fm = 0
            for fa, fb in zip(fake_feats, real_feats):
                for a, b in zip(fa, fb):
                    fm = fm + F.l1_loss(a.float(), b.float())
            lam = self.lambda_fm if self.lambda_fm != 'auto' else (g_loss / fm).detach()
            adv = adv + fm * lam
            g_loss = g_loss + adv
            losses['fm_loss'], losses['adv_loss'], losses['g_loss'] = fm, adv, g_loss

        self.optimizer.zero_grad(['autoencoder'])
        g_loss.backward()
        self._sync_grads()
        self.grad_norm = nn.utils.clip_grad_norm_(ae.parameters(), self.grad_clip_thresh)
        self.optimizer.step(['autoencoder'])
        return {'loss': {k: (v.detach() if torch.is_tensor(v) else v) for k, v in losses.items()}}
