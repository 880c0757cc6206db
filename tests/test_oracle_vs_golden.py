"""CPU: pins the oracle (oracle/*.py) against fixtures generated from the reference itself
(tests/golden/make_golden.py).  Tolerance: 1e-3 abs on fp32 outputs (north_star), VQ indices exact,
post-step VQ buffers 1e-5 relative-ish."""
import json
import os

import numpy as np
import pytest
import torch

from _util import GOLDEN, SMALL_TRAINER, fmap_digest, json_field, load_npz, small_task_cfg, t
from oracle import audio
from oracle.model import discriminator_forward, hifigan_generator, msmc_vqgan_forward
from oracle.step import OracleTrainer, lr_at, prepare_params
from oracle.vq import multi_head_quantize

import oracle.model as oracle_model

oracle_model.RESSTACK_DROPOUT = 0.0        # fixtures were generated with every nn.Dropout.p = 0
torch.set_num_threads(4)
TOL = 1e-3


def close(a, b, tol=TOL, rel=0.0):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b)
    bound = tol + rel * np.abs(b)
    assert (err <= bound).all(), 'max err %.3e (tol %.1e, rel %.1e)' % (err.max(), tol, rel)


# ------------------------------------------------------------------------------------ VQ
def _vq_cases():
    z = load_npz('vq_cases.npz')
    return z, json_field(z['cases'])


@pytest.mark.parametrize('idx', range(5))
def test_vq_matches_reference(idx):
    z, cases = _vq_cases()
    c = cases[idx]
    name, H = c['name'], c['H']
    heads = []
    for h in range(H):
        e = t(z['%s.init.embed.%d' % (name, h)]).clone()
        heads.append((e, torch.zeros(e.shape[1]), e.clone()))
    ln = t(z['%s.len' % name])
    for step in (0, 1):
        x = t(z['%s.s%d.x' % (name, step)]).clone().requires_grad_(True)
        q, d, ind = multi_head_quantize(x, ln, heads, True)
        want_ind = z['%s.s%d.ind' % (name, step)]
        if H == 1:
            ind = ind.squeeze(-1)
        assert np.array_equal(ind.numpy(), want_ind), 'indices must be bit-exact'
        close(q, z['%s.s%d.quant' % (name, step)], 1e-6)
        close(d, z['%s.s%d.diff' % (name, step)], 1e-6, 1e-6)
        (q.sum() * 0.5 + (d * torch.arange(d.numel()).view_as(d) / d.numel()).sum()).backward()
        close(x.grad, z['%s.s%d.grad_x' % (name, step)], 1e-6, 1e-6)
        for h in range(H):
            close(heads[h][0], z['%s.s%d.embed.%d' % (name, step, h)], 1e-5, 1e-5)
            close(heads[h][1], z['%s.s%d.cluster_size.%d' % (name, step, h)], 1e-6, 1e-6)
            close(heads[h][2], z['%s.s%d.embed_avg.%d' % (name, step, h)], 1e-6, 1e-6)
    x = t(z['%s.s0.x' % name])
    q, d, ind = multi_head_quantize(x, ln, heads, False)
    if H == 1:
        ind = ind.squeeze(-1)
    assert np.array_equal(ind.numpy(), z['%s.eval.ind' % name])
    close(q, z['%s.eval.quant' % name], 1e-6)


# ------------------------------------------------------------------------------------ front-ends
def test_frontends_match_reference():
    z = load_npz('frontends.npz')
    wav, wav2 = t(z['wav']), t(z['wav2'])
    for hop in (15, 60, 120):
        close(audio.mrd_spectrogram(wav, hop), z['mrd_image.%d' % hop], 1e-4)
    close(audio.mel_loss_spectrogram(wav, 2048, 300, 1200, 24000, 128), z['melloss.logmel'], 1e-4)
    close(audio.mel_loss(wav2, wav), z['melloss.value'], 1e-5)
    close(audio.mel_loss(wav2, wav, sample_rate=16000), z['melloss16.value'], 1e-5)
    r = audio.mr_stft_loss(wav2, wav)
    close(r['sc_loss'], z['mrstft.sc'], 1e-5)
    close(r['mag_loss'], z['mrstft.mag'], 1e-5)


def test_slaney_basis_properties():
    m = audio.slaney_mel_basis(24000, 2048, 128, 0, 12000)
    assert m.shape == (128, 1025) and m.dtype == np.float32
    assert (m >= 0).all() and (m.sum(1) > 0).all()
    peaks = m.argmax(1)
    assert (np.diff(peaks) > 0).all()                   # centre frequencies increase
    lin = peaks[:20].astype(np.float64) * 24000 / 2048   # below 1 kHz the Slaney scale is linear
    assert np.allclose(np.diff(lin), np.diff(lin)[0], atol=24000 / 2048 + 1e-6)


# ------------------------------------------------------------------------------------ modules
def _small():
    sd = load_npz('small_state.npz')
    return {k: t(v) for k, v in sd.items()}


def test_autoencoder_forward_matches_reference():
    z = load_npz('small_modules.npz')
    cfg = small_task_cfg()['autoencoder']
    P = prepare_params(_small())
    win = [tuple(int(v) for v in r) for r in z['windows']]
    out = msmc_vqgan_forward(P, cfg, t(z['batch.mel']), t(z['batch.mel_length']), warmup=False, window=win,
                             training=True)
    for i in range(2):
        assert np.array_equal(out['encoder_indices'][i].numpy(), z['ae.encoder_indices.%d' % i])
        assert np.array_equal(out['encoder_lengths'][i].numpy(), z['ae.encoder_lengths.%d' % i])
        close(out['encoder_outputs'][i], z['ae.encoder_outputs.%d' % i])
        close(out['encoder_diffs'][i], z['ae.encoder_diffs.%d' % i])
    close(out['mel_outputs'], z['ae.mel_outputs'])
    close(out['decoder_outputs'], z['ae.decoder_outputs'])
    close(out['decoder_diffs']['embed_loss_mse_1'], z['ae.embed_loss_mse_1'])
    for k, v in z.items():
        if k.startswith('ae.post.'):
            close(P[k[len('ae.post.'):]], v, 1e-5, 1e-5)


def test_generator_matches_reference():
    z = load_npz('small_modules.npz')
    P = prepare_params(_small())
    y = hifigan_generator(P, 'autoencoder.decoder', t(z['gen.in']), small_task_cfg()['autoencoder']['decoder_config'])
    close(y, z['gen.out'])


@pytest.mark.parametrize('tag', ['real', 'fake'])
def test_discriminator_matches_reference(tag):
    z = load_npz('small_modules.npz')
    # the module fixture ran D on the state *after* one autoencoder forward; D params are untouched by it
    P = prepare_params(_small())
    scores, fmaps = discriminator_forward(P, small_task_cfg()['discriminator'], t(z['disc.%s.in' % tag]))
    assert len(scores) == 4 and [len(f) for f in fmaps] == [6, 6, 5, 5]
    for i, s in enumerate(scores):
        close(s, z['disc.%s.score.%d' % (tag, i)])
    for i, fl in enumerate(fmaps):
        for j, f in enumerate(fl):
            assert list(f.shape) == z['disc.%s.fmap_shape.%d.%d' % (tag, i, j)].tolist()
            close(fmap_digest(f), z['disc.%s.fmap.%d.%d' % (tag, i, j)])
    if tag == 'real':
        close(fmaps[0][0], z['disc.real.fmap_full.0.0'])      # post-LeakyReLU (aliasing quirk)
        close(fmaps[2][1], z['disc.real.fmap_full.2.1'])      # pre-activation


# ------------------------------------------------------------------------------------ train steps
@pytest.mark.parametrize('tag,iteration', [('warm', 0), ('gan', 6)])
def test_train_step_matches_reference(tag, iteration):
    z = load_npz('small_steps.npz')
    tr = OracleTrainer(_small(), small_task_cfg(), SMALL_TRAINER)
    batch = {k[len('batch.'):]: t(v) for k, v in z.items() if k.startswith('batch.')}
    fw = [tuple(int(v) for v in r) for r in z['windows']]
    sw = [(s * 300, e * 300) for s, e in fw]
    keep = {}
    log = tr.train_step(batch, iteration, windows=(fw, sw), keep=keep)
    want = {k[len(tag) + 6:]: float(v) for k, v in z.items() if k.startswith(tag + '.loss.')}
    assert set(want) == set(log['loss']), (sorted(want), sorted(log['loss']))
    for k, v in want.items():
        assert abs(log['loss'][k] - v) <= TOL * max(1.0, abs(v)), (k, log['loss'][k], v)
    # gradients (post-clip for the autoencoder, as snapshotted at optimizer.step in the reference)
    for child in ('autoencoder', 'discriminator'):
        key = '%s.grad_names.%s' % (tag, child)
        if key not in z:
            continue
        names = json_field(z[key])
        l2 = z['%s.grad_l2.%s' % (tag, child)]
        got = keep['d_grads'] if child == 'discriminator' else \
            {k: tr.P[k].grad for k in names}
        assert set(names) == {k for k in got if got[k] is not None}
        for n, w in zip(names, l2):
            g = got[n].double().norm().item()
            assert abs(g - w) <= 2e-3 * max(w, 1e-3) + 1e-6, (n, g, w)
        for k, v in z.items():
            if k.startswith('%s.grad.%s.' % (tag, child)):
                n = k[len(tag) + 6:]
                close(got[n], v, 1e-5, 2e-3)
    for k, v in z.items():
        if k.startswith(tag + '.post.'):
            n = k[len(tag) + 6:]
            if n.endswith(('.embed', '.cluster_size', '.embed_avg')):
                close(tr.P[n], v, 1e-5, 1e-4)
            else:
                close(tr.P[n], v, 4.5e-4)       # one AdamW step moves a parameter by <= lr*(1+eps) = 2e-4


def test_predictor_step():
    """config #4: the oracle's predictor step against the reference's (tests/golden/small_predictor.npz)"""
    from oracle import predictor as op
    from _util import PREDICTOR_TRAINER, small_predictor_cfg
    z = load_npz('small_predictor.npz')
    P = {k[len('state.'):]: t(v).clone() for k, v in z.items() if k.startswith('state.')}
    for k, v in P.items():
        if not k.endswith('position.weight'):
            v.requires_grad_(True)
    P_ae = prepare_params(_small())
    batch = {k[len('batch.'):]: t(v) for k, v in z.items() if k.startswith('batch.')}
    losses, grads, out = op.predictor_step(P, small_predictor_cfg(), P_ae, small_task_cfg()['autoencoder'], batch,
                                           PREDICTOR_TRAINER['training_methods'], PREDICTOR_TRAINER['loss_weights'],
                                           PREDICTOR_TRAINER['lambda_dur'], PREDICTOR_TRAINER['grad_clip_thresh'])
    for i in range(2):
        close(out['feat'][i], z['fwd.feat.%d' % i], 1e-5)
        assert np.array_equal(out['feat_length'][i].numpy(), z['fwd.feat_length.%d' % i])
    close(out['duration'], z['fwd.duration'], 1e-5)
    want = {k[len('loss.'):]: float(v) for k, v in z.items() if k.startswith('loss.')}
    assert set(want) == set(losses)
    for k, v in want.items():
        assert abs(float(losses[k]) - v) <= 1e-4 * max(1.0, abs(v)), (k, float(losses[k]), v)
    names = json_field(z['grad_names'])
    assert set(names) == set(grads), set(names) ^ set(grads)
    for n, w in zip(names, z['grad_l2']):
        g = grads[n].double().norm().item()
        assert abs(g - w) <= 1e-3 * max(w, 1e-3) + 1e-6, (n, g, w)


def test_lr_schedule_matches_reference():
    with open(os.path.join(GOLDEN, 'schedule.json')) as f:
        s = json.load(f)
    for step, lr in zip(s['lr_steps'], s['lr_values']):
        got = lr_at(step, 2e-4, warmup_steps=200000, decay_scale=200000, decay_learning_rate=0.5,
                    final_learning_rate=1e-5)
        assert abs(got - lr) < 1e-12
