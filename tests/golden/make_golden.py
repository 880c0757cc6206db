#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the reference itself.

Runs ONLY in the build container (needs /root/reference); the GPU box and CI consume the
committed ``*.npz`` / ``*.json`` files.  Usage::

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Every fixture is data (inputs + the reference's outputs); no reference source is stored.
Fixtures
  vq_cases.npz          Quantize / MultiHeadQuantize (modules.py:24-67,137-151): outputs + EMA buffers
  frontends.npz         TorchSTFT('double') images, MelLoss log-mels + loss, MR-STFT losses
  small_state.npz       state_dict of the small model (SURVEY appendix C recipe)
  small_modules.npz     autoencoder forward outputs, discriminator scores / fmaps on that state
  small_steps.npz       one warm-up train_step and one GAN train_step (losses, gradients, VQ buffers)
  schedule.json         ExponentialDecayLRScheduler values; CSMSC state_dict key/shape list
"""
import copy
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from msmctts.networks.vqgantts.modules import MultiHeadQuantize, Quantize  # noqa: E402
from msmctts.tasks import build_task  # noqa: E402
from msmctts.trainers.criterions.stft_loss import MelLoss, MultiResolutionSTFTLoss  # noqa: E402
from msmctts.trainers.lr_schedulers.exponential_lr import ExponentialDecayLRScheduler  # noqa: E402
from msmctts.trainers.msmctts_trainer import VQGANTrainer  # noqa: E402
from msmctts.trainers.optimizers import build_optimizer  # noqa: E402
from msmctts.utils.audio import TorchSTFT  # noqa: E402
from msmctts.utils.config import Config  # noqa: E402

torch.set_num_threads(4)

SMALL_TASK = {
    '_name': 'MSMCTTS', '_mode': 'train_autoencoder',
    'autoencoder': {
        '_name': 'MSMCVQGAN', 'in_dim': 80, 'n_model_size': 32,
        'encoder_config': dict(downsample_scales=[1, 4], max_seq_len=64, n_layers=1, n_head=2, d_k=8, d_v=8,
                               d_inner=64, fft_conv1d_kernel=3, fft_conv1d_padding=1, dropout=0.0,
                               attn_dropout=0.0, fused_layernorm=False),
        'quantizer_config': dict(embedding_sizes=16, embedding_dims=32, n_heads=4,
                                 prior_config=dict(kernel_size=5, dilation_rate=1, n_layers=1),
                                 norm=False, dropout=0.0),
        'frame_decoder_config': dict(max_seq_len=64, n_layers=1, n_head=2, d_k=8, d_v=8, d_inner=64,
                                     fft_conv1d_kernel=3, fft_conv1d_padding=1, dropout=0.0,
                                     attn_dropout=0.0, fused_layernorm=False),
        'pred_mel': True,
        'decoder_config': dict(upsample_rates=[6, 5, 5, 2], upsample_kernel_sizes=[12, 11, 11, 4],
                               upsample_initial_channel=32, resblock_kernel_sizes=[3, 7, 11],
                               resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]]),
    },
    'discriminator': {
        '_name': 'UnivNetDiscriminator',
        'mrd_config': dict(hop_lengths=[15, 60], hidden_channels=[32, 32], domain='double', mel_scale=True,
                           sample_rate=24000),
        'mpd_config': dict(periods=[2, 3], channels=4, max_channels=16),
    },
}
SMALL_TRAINER = dict(_name='VQGANTrainer', grad_clip_thresh=1.0, warmup_steps=5, sample_lengths=2400,
                     lambda_vq=1, lambda_pr=0.1, lambda_frame=450, lambda_fm=2, lambda_stft=45)
OPTIM = {'_default': dict(_name='AdamW', learning_rate=2e-4, betas=[0.8, 0.99], eps=1e-8, weight_decay=0.0)}
DATASET = dict(_name='MelDataset', samplerate=24000, feature=['mel', 'wav'], dimension=[80, 1],
               frameshift=[300, 1], padding_value=[-4, 0])


def small_config():
    return Config({'id': 'golden_small', 'task': copy.deepcopy(SMALL_TASK), 'trainer': dict(SMALL_TRAINER),
                   'optimizer': copy.deepcopy(OPTIM), 'dataset': copy.deepcopy(DATASET),
                   'dataloader': {'batch_size': 3, 'num_workers': 0}})


def zero_dropout(module):
    for m in module.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0


def npy(t):
    return t.detach().cpu().numpy()


def small_batch(seed=7, B=3, T=24, lengths=(24, 17, 9), hop=300):
    g = torch.Generator().manual_seed(seed)
    mel = torch.randn(B, T, 80, generator=g)
    ln = torch.tensor(lengths, dtype=torch.int64)
    for i, l in enumerate(lengths):
        mel[i, l:] = -4.0
    wav = torch.rand(B, T * hop, 1, generator=g) * 2 - 1
    for i, l in enumerate(lengths):
        wav[i, l * hop:] = 0.0
    return {'mel': mel, 'mel_length': ln, 'wav': wav, 'wav_length': ln * hop}


# ---------------------------------------------------------------------------------------------
def gen_vq(out):
    cases = [  # name, B, T, D, H, K, lengths
        ('h1k16', 3, 17, 32, 1, 16, (17, 11, 4)),
        ('h4k16', 3, 17, 32, 4, 16, (17, 11, 4)),
        ('h8k64', 2, 33, 64, 8, 64, (33, 20)),
        ('h4k256', 2, 40, 64, 4, 256, (40, 23)),
        ('h2k512', 2, 24, 32, 2, 512, (24, 1)),
    ]
    meta = []
    for name, B, T, D, H, K, lens in cases:
        torch.manual_seed(sum(ord(c) for c in name) + 11)
        q = Quantize(D, K) if H == 1 else MultiHeadQuantize(D, K, H)
        heads = [q] if H == 1 else list(q.quantizers)
        x0 = torch.randn(B, T, D)
        x1 = torch.randn(B, T, D) * 0.7 + 0.1
        ln = torch.tensor(lens, dtype=torch.int64)
        for h, m in enumerate(heads):
            out['%s.init.embed.%d' % (name, h)] = npy(m.embed).copy()
        q.train()
        for step, x in enumerate((x0, x1)):          # two consecutive EMA updates ("step 1 and a later step")
            xi = x.clone().requires_grad_(True)
            qq, dd, ii = q(xi, ln, update=True)
            (qq.sum() * 0.5 + (dd * torch.arange(dd.numel()).view_as(dd) / dd.numel()).sum()).backward()
            out['%s.s%d.x' % (name, step)] = npy(x)
            out['%s.s%d.quant' % (name, step)] = npy(qq)
            out['%s.s%d.diff' % (name, step)] = npy(dd)
            out['%s.s%d.ind' % (name, step)] = npy(ii)
            out['%s.s%d.grad_x' % (name, step)] = npy(xi.grad)
            for h, m in enumerate(heads):
                out['%s.s%d.embed.%d' % (name, step, h)] = npy(m.embed).copy()
                out['%s.s%d.cluster_size.%d' % (name, step, h)] = npy(m.cluster_size).copy()
                out['%s.s%d.embed_avg.%d' % (name, step, h)] = npy(m.embed_avg).copy()
        q.eval()                                     # search only, no update
        qq, dd, ii = q(x0, ln, update=True)
        out['%s.eval.quant' % name] = npy(qq)
        out['%s.eval.ind' % name] = npy(ii)
        out['%s.len' % name] = npy(ln)
        meta.append(dict(name=name, B=B, T=T, D=D, H=H, K=K))
    out['cases'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)


def gen_frontends(out):
    g = torch.Generator().manual_seed(3)
    wav = (torch.rand(2, 2400, generator=g) * 2 - 1) * 0.8
    wav2 = (wav + 0.05 * torch.randn(2, 2400, generator=g)).clamp(-1, 1)
    out['wav'], out['wav2'] = npy(wav), npy(wav2)
    for hop in (15, 60, 120):
        st = TorchSTFT(fft_size=hop * 4, hop_size=hop, win_size=hop * 4, normalized=True, domain='double',
                       mel_scale=True, sample_rate=24000)
        mag, _ = st.transform(wav)
        out['mrd_image.%d' % hop] = npy(torch.stack(torch.chunk(mag, 2, dim=1), dim=1))
    ml = MelLoss(fft_size=2048, hop_size=300, win_size=1200, sample_rate=24000, num_mels=128)
    out['melloss.logmel'] = npy(ml.mel_spectrogram(wav))
    out['melloss.value'] = npy(ml(wav2, wav))
    ml16 = MelLoss(fft_size=1024, hop_size=200, win_size=800, sample_rate=16000, num_mels=128)
    out['melloss16.value'] = npy(ml16(wav2, wav))
    mr = MultiResolutionSTFTLoss()
    r = mr(wav2, wav)
    out['mrstft.sc'], out['mrstft.mag'] = npy(r['sc_loss']), npy(r['mag_loss'])


def fmap_digest(t):
    f = t.detach().reshape(-1).double()
    return np.concatenate([[f.mean().item(), f.abs().mean().item(), f.std().item(), float(f.numel())],
                           f[::7][:512].numpy()]).astype(np.float64)


def build_small(seed):
    torch.manual_seed(seed)
    cfg = small_config()
    task = build_task(cfg, mode='train')
    zero_dropout(task)
    # spread the VQ inputs / codebooks a little so that indices are not degenerate
    task.train()
    return cfg, task


def gen_small(state_out, mod_out, step_out):
    cfg, task = build_small(1234)
    sd0 = copy.deepcopy(task.state_dict())
    for k, v in sd0.items():
        state_out[k] = npy(v)
    batch = small_batch()
    for k, v in batch.items():
        mod_out['batch.' + k] = npy(v)
        step_out['batch.' + k] = npy(v)
    windows = [(3, 11), (0, 8), (1, 9)]
    mod_out['windows'] = np.asarray(windows, dtype=np.int64)

    # ---- module-level forward on a scratch copy (train mode => EMA update happens once)
    _, t2 = build_small(1234)
    t2.load_state_dict(sd0)
    o = t2.autoencoder(batch['mel'], batch['mel_length'], warmup=False, window=windows)
    mod_out['ae.mel_outputs'] = npy(o['mel_outputs'])
    mod_out['ae.decoder_outputs'] = npy(o['decoder_outputs'])
    for i in range(2):
        mod_out['ae.encoder_outputs.%d' % i] = npy(o['encoder_outputs'][i])
        mod_out['ae.encoder_indices.%d' % i] = npy(o['encoder_indices'][i])
        mod_out['ae.encoder_diffs.%d' % i] = npy(o['encoder_diffs'][i])
        mod_out['ae.encoder_lengths.%d' % i] = npy(o['encoder_lengths'][i])
    mod_out['ae.embed_loss_mse_1'] = npy(o['decoder_diffs']['embed_loss_mse_1'])
    for k, v in t2.state_dict().items():
        if k.endswith(('.embed', '.cluster_size', '.embed_avg')):
            mod_out['ae.post.' + k] = npy(v)
    # generator alone
    g = torch.Generator().manual_seed(5)
    gin = torch.randn(2, 32, 8, generator=g)
    mod_out['gen.in'] = npy(gin)
    mod_out['gen.out'] = npy(t2.autoencoder.decoder(gin))
    # discriminator on real-ish and fake-ish audio
    sw = [(s * 300, e * 300) for s, e in windows]
    real = torch.stack([batch['wav'][i, s:e] for i, (s, e) in enumerate(sw)], 0).squeeze(-1)
    fake = o['decoder_outputs'].detach().squeeze(-1)
    for tag, y in (('real', real), ('fake', fake)):
        scores, fmaps = t2.discriminator(y)
        mod_out['disc.%s.in' % tag] = npy(y)
        for i, s in enumerate(scores):
            mod_out['disc.%s.score.%d' % (tag, i)] = npy(s)
        for i, fl in enumerate(fmaps):
            for j, f in enumerate(fl):
                mod_out['disc.%s.fmap.%d.%d' % (tag, i, j)] = fmap_digest(f)
                mod_out['disc.%s.fmap_shape.%d.%d' % (tag, i, j)] = np.asarray(f.shape)
        if tag == 'real':
            mod_out['disc.real.fmap_full.0.0'] = npy(fmaps[0][0])
            mod_out['disc.real.fmap_full.2.1'] = npy(fmaps[2][1])

    # ---- train steps: A = warm-up step from sd0, B = GAN step from sd0 (fresh optimizer each)
    for tag, iteration in (('warm', 0), ('gan', 6)):
        cfg, tk = build_small(1234)
        tk.load_state_dict(sd0)
        tcfg = {k: v for k, v in SMALL_TRAINER.items() if k != '_name'}
        tr = VQGANTrainer(cfg, tk, num_gpus=0, rank=0, **tcfg)
        tr.optimizer = build_optimizer(tk, cfg.optimizer)
        snaps = {}
        real_step = tr.optimizer.step

        def spy(names=None, _snaps=snaps, _tk=tk, _real=real_step):
            key = names[0] if isinstance(names, (list, tuple)) else names
            _snaps[key] = {n: p.grad.detach().clone() for n, p in _tk.named_parameters()
                           if n.startswith(key + '.') and p.grad is not None}
            return _real(names)

        tr.optimizer.step = spy
        tr.random_select = lambda ml: (windows, sw)
        tk.zero_grad()
        tr.optimizer.zero_grad()
        log = tr.train_step({k: v.clone() for k, v in batch.items()}, iteration)
        for k, v in log['loss'].items():
            step_out['%s.loss.%s' % (tag, k)] = np.asarray(float(v), dtype=np.float64)
        for child, gd in snaps.items():
            names = sorted(gd)
            step_out['%s.grad_names.%s' % (tag, child)] = np.frombuffer(json.dumps(names).encode(), np.uint8)
            step_out['%s.grad_l2.%s' % (tag, child)] = np.asarray([gd[n].double().norm().item() for n in names])
            step_out['%s.grad_sum.%s' % (tag, child)] = np.asarray([gd[n].double().sum().item() for n in names])
            for n in names:
                if gd[n].numel() <= 2048 or n.endswith(('in_linear.weight', 'conv_post.weight_v')):
                    step_out['%s.grad.%s' % (tag, n)] = npy(gd[n])
        for k, v in tk.state_dict().items():
            if k.endswith(('.embed', '.cluster_size', '.embed_avg')):
                step_out['%s.post.%s' % (tag, k)] = npy(v)
            elif k.endswith(('in_linear.bias', 'mel_predictor.bias', 'conv_post.bias')):
                step_out['%s.post.%s' % (tag, k)] = npy(v)
    step_out['windows'] = np.asarray(windows, dtype=np.int64)


def gen_schedule(path):
    sch = ExponentialDecayLRScheduler(warmup_steps=200000, decay_scale=200000, decay_learning_rate=0.5,
                                      final_learning_rate=1e-5)
    steps = [0, 1, 199999, 200000, 200001, 300000, 400000, 800000, 1200000, 2000000]
    lrs = [max(1e-5, float(sch.get_scale(s)) * 2e-4) for s in steps]
    cfg = Config(os.path.join(_ref_shims.REFERENCE_ROOT, 'examples/csmsc/configs/msmc_vq_gan.yaml'))
    task = build_task(cfg, mode='train')
    keys = [[k, list(v.shape)] for k, v in task.state_dict().items()]
    n_param = {c: sum(p.numel() for p in m.parameters()) for c, m in task.named_children()}
    with open(path, 'w') as f:
        json.dump({'lr_steps': steps, 'lr_values': lrs, 'csmsc_state_dict': keys, 'csmsc_param_counts': n_param},
                  f, indent=0)


def main():
    random.seed(0)
    vq, fe, st, mo, sp = {}, {}, {}, {}, {}
    gen_vq(vq)
    gen_frontends(fe)
    gen_small(st, mo, sp)
    np.savez_compressed(os.path.join(HERE, 'vq_cases.npz'), **vq)
    np.savez_compressed(os.path.join(HERE, 'frontends.npz'), **fe)
    np.savez_compressed(os.path.join(HERE, 'small_state.npz'), **st)
    np.savez_compressed(os.path.join(HERE, 'small_modules.npz'), **mo)
    np.savez_compressed(os.path.join(HERE, 'small_steps.npz'), **sp)
    gen_schedule(os.path.join(HERE, 'schedule.json'))
    for n in sorted(os.listdir(HERE)):
        if n.endswith(('.npz', '.json')):
            print('%-22s %8.1f kB' % (n, os.path.getsize(os.path.join(HERE, n)) / 1024))


if __name__ == '__main__':
    main()
