"""Oracle: one ``VQGANTrainer.train_step`` on CPU (plain PyTorch fp32).  TEST INFRASTRUCTURE ONLY.

Restates
  * ``QuantizerLoss.forward``       reference msmctts/trainers/msmctts_trainer.py:39-71
  * ``VQGANTrainer.train_step``     reference msmctts/trainers/msmctts_trainer.py:115-209
  * ``VQGANTrainer.random_select``  reference msmctts/trainers/msmctts_trainer.py:211-219
  * optimizer bundle / LR schedule  reference msmctts/trainers/optimizers/__init__.py:24-78,
                                    msmctts/trainers/lr_schedulers/exponential_lr.py:16-30
  * loop-level zero_grad            reference msmctts/trainers/base_trainer.py:84-85
"""
import random

import numpy as np
import torch
import torch.nn.functional as F

from . import audio
from .model import discriminator_forward, msmc_vqgan_forward, pad_mask

BUFFER_SUFFIXES = ('.embed', '.cluster_size', '.embed_avg')


def is_buffer(name):
    return name.endswith(BUFFER_SUFFIXES)


def is_frozen(name):
    return name.endswith('.position.weight')        # nn.Embedding.from_pretrained(freeze=True)


def prepare_params(state_dict):
    """Clone a reference-keyed state dict into leaf tensors with the reference's requires_grad flags."""
    P = {}
    for k, v in state_dict.items():
        t = torch.as_tensor(v).detach().clone().float() if not torch.is_tensor(v) else v.detach().clone().float().cpu()
        t.requires_grad_(not (is_buffer(k) or is_frozen(k)))
        P[k] = t
    return P


def quantizer_loss(out, lambda_vq=1, lambda_pr=1):
    """msmctts_trainer.py:45-71; does not mutate ``out`` (the reference masks the diffs in place)."""
    loss = {'vq_loss': 0}
    for i, term in enumerate(out['encoder_diffs']):
        length = out['encoder_lengths'][i]
        term = term.masked_fill(pad_mask(length, term.shape[1]).unsqueeze(-1), 0)
        term = term.sum() / sum(length) / term.shape[2]
        loss['latent_loss_%d_0' % i] = term
        loss['vq_loss'] = loss['vq_loss'] + lambda_vq * term
    dd = out.get('decoder_diffs')
    if isinstance(dd, dict):
        dd = dict(dd)
        loss['vq_loss'] = loss['vq_loss'] + lambda_pr * dd.pop('total_loss')
        loss.update(dd)
    return loss


def random_select(mel_length, frame_lengths, frameshift, rng=random):
    """msmctts_trainer.py:211-219 (python global RNG by default)."""
    fw, sw = [], []
    for i in range(mel_length.shape[0]):
        start = rng.randrange(max(1, int(mel_length[i]) - frame_lengths))
        end = start + frame_lengths
        fw.append((start, end))
        sw.append((start * frameshift, end * frameshift))
    return fw, sw


def lr_at(step, base_lr, warmup_steps=50000, decay_scale=50000, decay_learning_rate=0.5,
          final_learning_rate=1e-5):
    """exponential_lr.py:16-30."""
    scale = np.power(decay_learning_rate, (step - warmup_steps) / decay_scale) if step >= warmup_steps else 1.0
    return max(final_learning_rate, scale * base_lr)


class OracleTrainer(object):
    """Holds parameters + per-child AdamW and executes reference-equivalent steps."""

    def __init__(self, state_dict, task_cfg, trainer_cfg, optim_cfg=None, frameshift=300,
                 sample_rate=24000, training=True):
        self.P = prepare_params(state_dict)
        self.acfg = task_cfg['autoencoder']
        self.dcfg = task_cfg.get('discriminator')
        t = dict(trainer_cfg)
        self.warmup_steps = t.get('warmup_steps', 0)
        self.lambda_frame = t.get('lambda_frame', 1.0)
        self.grad_clip_thresh = t.get('grad_clip_thresh', 1.0)
        self.sample_lengths = t.get('sample_lengths', 24000)
        self.lambda_vq, self.lambda_pr = t.get('lambda_vq', 1), t.get('lambda_pr', 1)
        self.lambda_fm, self.lambda_stft = t.get('lambda_fm', 2), t.get('lambda_stft', 45)
        self.stft_loss_func = t.get('stft_loss_func', 'mel_loss')
        self.stft_loss_config = t.get('stft_loss_config')
        self.frameshift, self.sample_rate = frameshift, sample_rate
        self.frame_lengths = -1 if self.sample_lengths == -1 else self.sample_lengths // frameshift
        self.training = training
        o = optim_cfg or {'learning_rate': 2e-4, 'betas': [0.8, 0.99], 'eps': 1e-8, 'weight_decay': 0.0}
        self.base_lr = o['learning_rate']
        self.opt = {}
        for child in ('autoencoder', 'discriminator'):
            ps = [p for k, p in self.P.items() if k.startswith(child + '.') and not is_buffer(k)]
            if ps:
                # the reference hands *all* module.parameters() (frozen tables included) to AdamW
                self.opt[child] = torch.optim.AdamW(ps, o['learning_rate'], tuple(o['betas']), o['eps'],
                                                    o['weight_decay'])

    def child_params(self, child, trainable_only=True):
        return [p for k, p in self.P.items() if k.startswith(child + '.') and not is_buffer(k)
                and (p.requires_grad or not trainable_only)]

    def zero_all(self):
        for p in self.P.values():
            p.grad = None

    def stft_criterion(self, predict, target):
        if self.stft_loss_func == 'mel_loss':
            kw = dict(sample_rate=self.sample_rate)
            if self.stft_loss_config:
                kw.update(self.stft_loss_config)
            return audio.mel_loss(predict, target, **kw)
        return audio.mr_stft_loss(predict, target, **(self.stft_loss_config or {}))

    def train_step(self, batch, iteration, windows=None, keep=None):
        """Returns {'loss': {...python floats...}}.  ``keep`` (dict) receives intermediate tensors."""
        P = self.P
        self.zero_all()                                  # base_trainer.py:84-85
        losses = {}
        mel, mel_length = batch['mel'], batch['mel_length']
        wav = batch['wav']
        if iteration < self.warmup_steps:
            out = msmc_vqgan_forward(P, self.acfg, mel, mel_length, warmup=True, training=self.training)
        else:
            if windows is None:
                windows = random_select(mel_length, self.frame_lengths, self.frameshift)
            fw, sw = windows
            target = torch.stack([wav[i, s:e] for i, (s, e) in enumerate(sw)], dim=0)
            out = msmc_vqgan_forward(P, self.acfg, mel, mel_length, warmup=False, window=fw,
                                     training=self.training)
        vq = quantizer_loss(out, self.lambda_vq, self.lambda_pr)
        losses.update(vq)
        g_loss = vq['vq_loss']
        if 'mel_outputs' in out:
            ml = F.mse_loss(mel, out['mel_outputs'], reduction='none')
            ml = ml.masked_fill(pad_mask(mel_length, ml.shape[1]).unsqueeze(-1), 0)
            ml = ml.sum() / sum(mel_length) / ml.shape[2]
            losses['frame_loss'] = ml.item()
            g_loss = g_loss + self.lambda_frame * ml
        if keep is not None:
            keep['autoencoder_out'] = out
        if iteration > self.warmup_steps:
            predict = out['decoder_outputs'].squeeze(-1)
            target = target.squeeze(-1)
            st = self.stft_criterion(predict, target)
            if isinstance(st, dict):
                tot = 0
                for n, term in st.items():
                    tot = tot + term
                    losses[n] = term.item()
                st = tot
            losses['stft_loss'] = st.item()
            g_loss = g_loss + self.lambda_stft * st
            # ---- discriminator step (msmctts_trainer.py:161-179)
            fs, _ = discriminator_forward(P, self.dcfg, predict.detach())
            rs, _ = discriminator_forward(P, self.dcfg, target)
            d_real = sum(F.mse_loss(r, torch.ones_like(r)) for r in rs)
            d_fake = sum(F.mse_loss(f, torch.zeros_like(f)) for f in fs)
            d_loss = d_real + d_fake
            losses['d_loss_real'], losses['d_loss_fake'], losses['d_loss'] = \
                d_real.item(), d_fake.item(), d_loss.item()
            self.opt['discriminator'].zero_grad()
            d_loss.backward()
            if keep is not None:
                keep['d_grads'] = {k: p.grad.clone() for k, p in P.items()
                                   if k.startswith('discriminator.') and p.grad is not None}
            self.opt['discriminator'].step()
            # ---- generator step (msmctts_trainer.py:181-201); D already updated
            fs, ff = discriminator_forward(P, self.dcfg, predict)
            rs, rf = discriminator_forward(P, self.dcfg, target)
            adv = sum(F.mse_loss(f, torch.ones_like(f)) for f in fs)
            fm = 0
            for a, b in zip(ff, rf):
                for x, y in zip(a, b):
                    fm = fm + F.l1_loss(x, y)
            lam = self.lambda_fm if self.lambda_fm != 'auto' else (g_loss / fm).detach()
            adv = adv + fm * lam
            g_loss = g_loss + adv
            losses['fm_loss'], losses['adv_loss'], losses['g_loss'] = fm.item(), adv.item(), g_loss.item()
        self.opt['autoencoder'].zero_grad()
        g_loss.backward()
        if keep is not None:
            keep['g_grads'] = {k: p.grad.clone() for k, p in P.items()
                               if k.startswith('autoencoder.') and p.grad is not None}
        grad_norm = torch.nn.utils.clip_grad_norm_(self.child_params('autoencoder', False),
                                                   self.grad_clip_thresh)
        if keep is not None:
            keep['grad_norm'] = float(grad_norm)
        self.opt['autoencoder'].step()
        out_losses = {k: (v.item() if torch.is_tensor(v) else float(v)) for k, v in losses.items()}
        return {'loss': out_losses}

    def set_lr(self, step, sched_cfg):
        lr = lr_at(step, self.base_lr, **sched_cfg)
        for o in self.opt.values():
            for g in o.param_groups:
                g['lr'] = lr
        return lr
