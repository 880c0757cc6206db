"""Parity checks shared by the CPU (kernel interpreter) and GPU (real gfx950 library) test modules.

Every check drives the PRODUCT path (msmctts_amd modules -> ctypes -> C ABI) and compares it with
the golden fixtures generated from the reference and/or with the oracle on the same seeded inputs.
Tolerances: fp32 outputs 1e-3 abs (north_star), VQ indices bit-exact, post-step VQ buffers 1e-5.
"""
import copy
import random

import numpy as np
import torch

from _util import PREDICTOR_TRAINER, SMALL_TRAINER, fmap_digest, json_field, load_npz, small_predictor_cfg, small_task_cfg, t

TOL = 1e-3


def close(a, b, tol=TOL, rel=0.0, what=''):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    assert (err <= tol + rel * np.abs(b)).all(), '%s: max err %.3e (tol %.1e rel %.1e)' % (what, err.max(), tol, rel)


def small_config():
    from msmctts_amd.utils.config import Config
    task = small_task_cfg()
    task['_name'] = 'MSMCTTS'
    return Config({'id': 'small', 'task': task, 'trainer': dict(SMALL_TRAINER, _name='VQGANTrainer'),
                   'optimizer': {'_default': dict(_name='AdamW', learning_rate=2e-4, betas=[0.8, 0.99], eps=1e-8,
                                                  weight_decay=0.0)},
                   'dataset': dict(samplerate=24000, feature=['mel', 'wav'], frameshift=[300, 1])})


def build_small(device):
    from msmctts_amd.tasks import build_task
    cfg = small_config()
    task = build_task(cfg, mode='train')
    sd = {k: t(v) for k, v in load_npz('small_state.npz').items()}
    task.load_state_dict(sd)
    for m in task.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return cfg, task.to(device).train()


# ------------------------------------------------------------------------------------------------
def check_vq_cases(device):
    from msmctts_amd.networks.vqgantts.modules import MultiHeadQuantize, Quantize
    z = load_npz('vq_cases.npz')
    for c in json_field(z['cases']):
        name, H, K, D = c['name'], c['H'], c['K'], c['D']
        q = Quantize(D, K) if H == 1 else MultiHeadQuantize(D, K, H)
        heads = [q] if H == 1 else list(q.quantizers)
        for h, m in enumerate(heads):
            e = t(z['%s.init.embed.%d' % (name, h)])
            m.embed.copy_(e)
            m.embed_avg.copy_(e)
            m.cluster_size.zero_()
        q = q.to(device).train()
        heads = [q] if H == 1 else list(q.quantizers)
        ln = t(z['%s.len' % name]).to(device)
        for step in (0, 1):
            x = t(z['%s.s%d.x' % (name, step)]).to(device).requires_grad_(True)
            qq, dd, ii = q(x, ln, update=True)
            assert np.array_equal(ii.cpu().numpy(), z['%s.s%d.ind' % (name, step)]), (name, step)
            scale = float(np.abs(z['%s.s%d.quant' % (name, step)]).max())
            close(qq, z['%s.s%d.quant' % (name, step)], 5e-6 * max(1.0, scale), what=name + ' quant')
            close(dd, z['%s.s%d.diff' % (name, step)], 5e-6 * max(1.0, scale * scale), 1e-5, what=name + ' diff')
            w = (torch.arange(dd.numel(), device=device).view_as(dd) / dd.numel())
            (qq.sum() * 0.5 + (dd * w).sum()).backward()
            close(x.grad, z['%s.s%d.grad_x' % (name, step)], 5e-6 * max(1.0, scale), 1e-5, what=name + ' grad')
            for h, m in enumerate(heads):
                ref = z['%s.s%d.embed.%d' % (name, step, h)]
                close(m.embed, ref, 1e-5 * max(1.0, float(np.abs(ref).max())), 1e-5, what=name + ' embed')
                close(m.cluster_size, z['%s.s%d.cluster_size.%d' % (name, step, h)], 1e-6, 1e-6)
                close(m.embed_avg, z['%s.s%d.embed_avg.%d' % (name, step, h)], 1e-5, 1e-6)
        q.eval()
        qq, dd, ii = q(t(z['%s.s0.x' % name]).to(device), ln, update=True)
        assert np.array_equal(ii.cpu().numpy(), z['%s.eval.ind' % name])
        # sort=True (reference modules.py:62-65): every codeword of every frame, nearest first; [.., K] / [.., K, H]
        _, _, rank = q(t(z['%s.s0.x' % name]).to(device), ln, update=True, sort=True)
        assert tuple(rank.shape) == tuple(ii.shape[:2]) + ((K,) if H == 1 else (K, H))
        kdim = -1 if H == 1 else -2
        assert torch.equal(rank.select(kdim, 0).cpu(), ii.cpu())
        assert torch.equal(rank.sort(dim=kdim)[0].cpu(),
                           torch.arange(K).view((K,) if H == 1 else (K, 1)).expand_as(rank))


def check_state_dict_surface():
    import json
    import os
    from _util import GOLDEN
    from msmctts_amd.tasks import build_task
    from msmctts_amd.utils.config import Config
    with open(os.path.join(GOLDEN, 'schedule.json')) as f:
        ref = json.load(f)
    from msmctts_amd.configs import csmsc_config
    task = build_task(Config(csmsc_config(embedding_sizes=64)), mode='train')
    mine = [[k, list(v.shape)] for k, v in task.state_dict().items()]
    assert mine == ref['csmsc_state_dict']
    counts = {c: sum(p.numel() for p in m.parameters()) for c, m in task.named_children()}
    assert counts == ref['csmsc_param_counts']


def check_modules(device):
    z = load_npz('small_modules.npz')
    cfg, task = build_small(device)
    win = [tuple(int(v) for v in r) for r in z['windows']]
    out = task.autoencoder(t(z['batch.mel']).to(device), t(z['batch.mel_length']).to(device), warmup=False,
                           window=win)
    for i in range(2):
        assert np.array_equal(out['encoder_indices'][i].cpu().numpy(), z['ae.encoder_indices.%d' % i])
        close(out['encoder_outputs'][i], z['ae.encoder_outputs.%d' % i], what='enc')
        close(out['encoder_diffs'][i], z['ae.encoder_diffs.%d' % i], what='diffs')
    close(out['mel_outputs'], z['ae.mel_outputs'], what='mel')
    close(out['decoder_outputs'], z['ae.decoder_outputs'], what='wav')
    close(out['decoder_diffs']['embed_loss_mse_1'], z['ae.embed_loss_mse_1'])
    sd = task.state_dict()
    for k, v in z.items():
        if k.startswith('ae.post.'):
            close(sd[k[len('ae.post.'):]], v, 1e-5, 1e-5, what=k)
    close(task.autoencoder.decoder(t(z['gen.in']).to(device)), z['gen.out'], what='gen')
    for tag in ('real', 'fake'):
        scores, fmaps = task.discriminator(t(z['disc.%s.in' % tag]).to(device))
        assert len(scores) == 4 and [len(f) for f in fmaps] == [6, 6, 5, 5]
        for i, s in enumerate(scores):
            close(s, z['disc.%s.score.%d' % (tag, i)], what='score')
        for i, fl in enumerate(fmaps):
            for j, f in enumerate(fl):
                assert list(f.shape) == z['disc.%s.fmap_shape.%d.%d' % (tag, i, j)].tolist()
                close(fmap_digest(f), z['disc.%s.fmap.%d.%d' % (tag, i, j)], what='fmap %d %d' % (i, j))


def check_train_steps(device, arm_reducer=False):
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.trainers.optimizers import build_optimizer
    z = load_npz('small_steps.npz')
    for tag, iteration in (('warm', 0), ('gan', 6)):
        cfg, task = build_small(device)
        tr = build_trainer(cfg, task, num_gpus=0, rank=0)
        tr.model = task
        if arm_reducer:      # data-parallel plumbing (hooks, buckets, collectives) with whatever group is initialised
            from msmctts_amd.distributed.distributed import apply_gradient_allreduce
            apply_gradient_allreduce(task)
        tr.optimizer = build_optimizer(task, cfg.optimizer)
        fw = [tuple(int(v) for v in r) for r in z['windows']]
        sw = [(s * 300, e * 300) for s, e in fw]
        tr.random_select = lambda ml: (fw, sw)
        batch = {k[len('batch.'):]: t(v).to(device) for k, v in z.items() if k.startswith('batch.')}
        snaps = {}
        real_step = tr.optimizer.step

        def spy(names=None):
            key = names[0] if isinstance(names, (list, tuple)) else names
            snaps[key] = {n: p.grad.detach().clone() for n, p in task.named_parameters()
                          if n.startswith(key + '.') and p.grad is not None}
            return real_step(names)

        real_clip_step = tr.optimizer.clip_and_step

        def clip_spy(name, max_norm):           # the fused clip + AdamW: gradients are clipped in place, as clip_grad_norm_ does
            out = real_clip_step(name, max_norm)
            snaps[name] = {n: p.grad.detach().clone() for n, p in task.named_parameters()
                           if n.startswith(name + '.') and p.grad is not None}
            return out
        tr.optimizer.step = spy
        tr.optimizer.clip_and_step = clip_spy
        task.zero_grad()
        log = tr.train_step(batch, iteration)
        want = {k[len(tag) + 6:]: float(v) for k, v in z.items() if k.startswith(tag + '.loss.')}
        assert set(want) == set(log['loss']), (sorted(want), sorted(log['loss']))
        for k, v in want.items():
            got = float(log['loss'][k])
            assert abs(got - v) <= TOL * max(1.0, abs(v)), (tag, k, got, v)
        for child, gd in snaps.items():
            names = json_field(z['%s.grad_names.%s' % (tag, child)])
            assert set(names) == set(gd), (set(names) ^ set(gd))
            for n, w in zip(names, z['%s.grad_l2.%s' % (tag, child)]):
                g = gd[n].double().norm().item()
                assert abs(g - w) <= 2e-3 * max(w, 1e-3) + 1e-6, (tag, n, g, w)
            for k, v in z.items():
                if k.startswith('%s.grad.%s.' % (tag, child)):
                    close(gd[k[len(tag) + 6:]], v, 1e-5, 2e-3, what=k)
        sd = task.state_dict()
        for k, v in z.items():
            if k.startswith(tag + '.post.'):
                n = k[len(tag) + 6:]
                if n.endswith(('.embed', '.cluster_size', '.embed_avg')):
                    close(sd[n], v, 1e-5, 1e-4, what=n)
                else:
                    close(sd[n], v, 4.5e-4, what=n)


def check_predictor_step(device):
    """BASELINE config #4: one PredictorTrainer.train_step of the product against the reference's own step
    (tests/golden/small_predictor.npz): forward predictions, durations, every loss, every clipped gradient norm, and the
    spot-checked post-step parameters."""
    from msmctts_amd.tasks import build_task
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.trainers.optimizers import build_optimizer
    from msmctts_amd.utils.config import Config
    z = load_npz('small_predictor.npz')
    cfg = Config({'id': 'small_predictor', 'task': {'_name': 'MSMCTTS', '_mode': 'train_predictor', 'predictor': small_predictor_cfg()},
                  'trainer': dict(PREDICTOR_TRAINER, _name='PredictorTrainer'),
                  'optimizer': {'_default': dict(_name='Adam', learning_rate=2e-4, betas=[0.9, 0.98], eps=1e-9, weight_decay=0)},
                  'dataset': dict(samplerate=24000, feature=['mel', 'wav'], frameshift=[300, 1])})
    task = build_task(cfg, mode='train')
    task.load_state_dict({k[len('state.'):]: t(v) for k, v in z.items() if k.startswith('state.')})
    task = task.to(device).train()
    _, atask = build_small(device)
    batch = {k[len('batch.'):]: t(v).to(device) for k, v in z.items() if k.startswith('batch.')}
    # forward alone (teacher-forced stage features), like the fixture
    atask.autoencoder.eval()
    with torch.no_grad():
        qs = atask.autoencoder.analysis(batch['mel'], batch['mel_length'].int())
        fo = task.predictor(text=batch['text'], text_length=batch['text_length'], dur=batch['dur'],
                            feat=[f.float() for f in qs['quantizer_outputs']], feat_length=qs['quantizer_lengths'])
    for i in range(2):
        assert np.array_equal(qs['quantizer_indices'][i].cpu().numpy(), z['ae.quantizer_indices.%d' % i])
        close(qs['quantizer_outputs'][i], z['ae.quantizer_outputs.%d' % i], 1e-5, what='analysis %d' % i)
        close(fo['feat'][i], z['fwd.feat.%d' % i], what='feat %d' % i)
        assert np.array_equal(fo['feat_length'][i].cpu().numpy(), z['fwd.feat_length.%d' % i])
    close(fo['duration'], z['fwd.duration'], what='duration')
    # one training step
    tr = build_trainer(cfg, task, num_gpus=0, rank=0)
    tr.autoencoder = atask.autoencoder
    tr.optimizer = build_optimizer(task, cfg.optimizer)
    snaps = {}
    real_step = tr.optimizer.step

    def spy(names=None):
        snaps['predictor'] = {n: p.grad.detach().clone() for n, p in task.named_parameters() if p.grad is not None}
        return real_step(names)
    tr.optimizer.step = spy
    if hasattr(tr.optimizer, 'clip_and_step'):
        real_clip_step = tr.optimizer.clip_and_step

        def clip_spy(name, max_norm):
            out = real_clip_step(name, max_norm)
            snaps[name] = {n: p.grad.detach().clone() for n, p in task.named_parameters() if p.grad is not None}
            return out
        tr.optimizer.clip_and_step = clip_spy
    log = tr.train_step({k: v.clone() for k, v in batch.items()}, 0)
    want = {k[len('loss.'):]: float(v) for k, v in z.items() if k.startswith('loss.')}
    assert set(want) == set(log['loss']), (sorted(want), sorted(log['loss']))
    for k, v in want.items():
        got = float(log['loss'][k])
        assert abs(got - v) <= TOL * max(1.0, abs(v)), (k, got, v)
    names = json_field(z['grad_names'])
    assert set(names) == set(snaps['predictor']), set(names) ^ set(snaps['predictor'])
    for n, w in zip(names, z['grad_l2']):
        g = snaps['predictor'][n].double().norm().item()
        assert abs(g - w) <= 2e-3 * max(w, 1e-3) + 1e-6, (n, g, w)
    sd = task.state_dict()
    for k, v in z.items():
        if k.startswith('post.'):
            close(sd[k[len('post.'):]], v, 1e-5, what=k)


def check_norm_kernels(dev, dtype, tol):
    """csrc/norm.hip against the stock operator chains they replace (forward and every gradient), ragged row counts,
    row mask; dropout: the backward pass regenerates exactly the forward's mask and the keep rate is right"""
    from msmctts_amd.hip import norm
    torch.manual_seed(0)
    for N, C in ((37, 256), (5, 96), (130, 32)):
        x = torch.randn(N, C, device=dev).to(dtype).requires_grad_(True)
        r = torch.randn(N, C, device=dev).to(dtype).requires_grad_(True)
        gm = (torch.rand(C, device=dev) + 0.5).requires_grad_(True)
        bt = torch.randn(C, device=dev).requires_grad_(True)
        keep = (torch.rand(N, device=dev) > 0.3).to(torch.uint8)
        y = norm.add_layer_norm(x, r, gm, bt, keep_row=keep)
        go = torch.randn(N, C, device=dev)
        (y.float() * go).sum().backward()
        xr, rr = x.detach().float().requires_grad_(True), r.detach().float().requires_grad_(True)
        gr, br = gm.detach().clone().requires_grad_(True), bt.detach().clone().requires_grad_(True)
        yr = torch.nn.functional.layer_norm(xr + rr, (C,), gr, br) * keep.float().unsqueeze(1)
        (yr * (go.to(dtype).float() if dtype != torch.float32 else go)).sum().backward()
        scale = lambda t: max(1.0, float(t.abs().max()))
        assert (y.float() - yr).abs().max() <= tol * scale(yr)
        assert (x.grad.float() - xr.grad).abs().max() <= tol * scale(xr.grad)
        assert (r.grad.float() - rr.grad).abs().max() <= tol * scale(rr.grad)
        assert (gm.grad - gr.grad).abs().max() <= tol * scale(gr.grad) * (4 if dtype != torch.float32 else 1)
        assert (bt.grad - br.grad).abs().max() <= tol * scale(br.grad) * (4 if dtype != torch.float32 else 1)
    # gate and tanh
    x = torch.randn(23, 2 * 48, device=dev).to(dtype).requires_grad_(True)
    y = norm.gate(x)
    go = torch.randn(23, 48, device=dev)
    (y.float() * go).sum().backward()
    xr = x.detach().float().requires_grad_(True)
    yr = torch.tanh(xr[:, :48]) * torch.sigmoid(xr[:, 48:])
    (yr * go).sum().backward()
    assert (y.float() - yr).abs().max() <= tol and (x.grad.float() - xr.grad).abs().max() <= tol * 4
    x = torch.randn(1000, device=dev).to(dtype).requires_grad_(True)
    y = norm.tanh(x)
    y.float().sum().backward()
    assert (y.float() - torch.tanh(x.detach().float())).abs().max() <= tol
    assert (x.grad.float() - (1 - torch.tanh(x.detach().float()) ** 2)).abs().max() <= tol * 2
    # dropout: same mask in both passes, keep rate ~ 1 - p, fresh mask after advance_seed
    p = 0.25
    x = torch.ones(64, 256, device=dev).to(dtype).requires_grad_(True)
    one, zero = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    salt = norm.new_salt()
    y = norm.gate(torch.full((64, 512), 3.0, device=dev).to(dtype).requires_grad_(True), p_drop=p, salt=salt)
    kept = (y != 0).float().mean().item()
    assert abs(kept - (1 - p)) < 0.03, kept
    xg = torch.full((64, 512), 3.0, device=dev).to(dtype).requires_grad_(True)
    yg = norm.gate(xg, p_drop=p, salt=salt)
    yg.float().sum().backward()
    assert torch.equal(yg != 0, y != 0)                                   # same seed, same salt: same mask
    assert torch.equal(xg.grad[:, :256] != 0, yg != 0)                    # backward regenerated it
    norm.advance_seed(xg.device)
    y2 = norm.gate(xg.detach(), p_drop=p, salt=salt)
    assert not torch.equal(y2 != 0, y != 0)


def check_fc_add_ln(dev):
    """The attention sub-layer's tail in one launch (csrc/norm.hip fc_add_ln_fwd_kernel behind hip/convnet.py hip_conv_add_ln) against
    (a) the stock fp32 chain layer_norm(linear(a) + res) * keep with every gradient, (b) the two-launch chain on the same bf16
    inputs: identical dropout masks, outputs within bf16 rounding of each other (the sums run in another order); ragged row
    counts, both tile counts (C <= 256 and C = 384), a shape the kernel refuses (falls back), and the launch count"""
    import torch.nn as nn
    from msmctts_amd.hip import convnet, lib, norm
    torch.manual_seed(3)
    dt = torch.bfloat16
    L = lib.get()
    for (B, T, K, C, p) in ((3, 37, 128, 256, 0.0), (2, 21, 64, 384, 0.0), (1, 16, 32, 36, 0.0), (3, 50, 128, 256, 0.2)):
        fc = nn.Linear(K, C).to(dev)
        ln = nn.LayerNorm(C).to(dev)
        with torch.no_grad():
            ln.weight.add_(torch.randn(C, device=dev) * 0.2)
            ln.bias.add_(torch.randn(C, device=dev) * 0.2)
        layer = convnet.ConvLayer(fc, 'conv', (1, 1), plain=True)
        bank = convnet.ConvBank([layer])
        a0 = torch.randn(B, 1, T, K, device=dev).to(dt)
        r0 = torch.randn(B, T, C, device=dev).to(dt)
        keep = (torch.rand(B * T, device=dev) > 0.25).to(torch.uint8)
        go = torch.randn(B, T, C, device=dev).to(dt).float()
        salt = norm.new_salt()

        def run(fused):
            saved, convnet.FC_LN_FUSE = convnet.FC_LN_FUSE, fused
            try:
                for q in list(fc.parameters()) + list(ln.parameters()):
                    q.grad = None
                bank.prepare(dt)
                a, r = a0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
                before = L.msmc_conv_launch_count()
                y = convnet.hip_conv_add_ln(bank, layer, a, r, ln.weight, ln.bias, keep_row=keep, p_drop=p, salt=salt, eps=ln.eps)
                launched = L.msmc_conv_launch_count() - before
                (y.float() * go).sum().backward()
                return y.detach().float(), a.grad.float(), r.grad.float(), [q.grad.clone() for q in (fc.weight, fc.bias, ln.weight, ln.bias)], launched
            finally:
                convnet.FC_LN_FUSE = saved
        yf, af, rf, pf, n_f = run(True)
        yu, au, ru, pu, n_u = run(False)
        assert convnet.fc_ln_fusable(layer, a0) and n_f == 0 and n_u >= 1, (n_f, n_u)      # (the projection left the conv launches; a first call of a shape also times its candidates)
        assert torch.equal(yf == 0, yu == 0)
        scale = lambda t: max(1.0, float(t.abs().max()))
        close(yf, yu, 0.04 * scale(yu), what='fused vs two launches')
        for g, w in zip([af, rf] + pf, [au, ru] + pu):
            close(g, w, 0.03 * scale(w), what='fused vs two launches: gradients')
        if p == 0.0:
            ar, rr = a0.float().requires_grad_(True), r0.float().requires_grad_(True)
            ref_fc, ref_ln = nn.Linear(K, C).to(dev), nn.LayerNorm(C).to(dev)
            with torch.no_grad():
                ref_fc.weight.copy_(fc.weight.to(dt).float()); ref_fc.bias.copy_(fc.bias)
                ref_ln.weight.copy_(ln.weight); ref_ln.bias.copy_(ln.bias)
            yr = ref_ln(ref_fc(ar).squeeze(1) + rr) * keep.float().view(B, T, 1)
            (yr * go).sum().backward()
            close(yf, yr, 0.03 * scale(yr), what='fused vs stock fp32 chain')
            close(af, ar.grad, 0.03 * scale(ar.grad), what='d a')
            close(rf, rr.grad, 0.03 * scale(rr.grad), what='d res')
            for g, w in zip(pf, (ref_fc.weight.grad, ref_fc.bias.grad, ref_ln.weight.grad, ref_ln.bias.grad)):
                close(g, w, 0.03 * scale(w), what='parameter gradients')
        else:
            kept = float((yf[keep.view(B, T).bool()] != 0).float().mean())
            assert kept > 0.99                                   # (dropout acts before the residual: every live row is dense)
    # shapes the kernel does not take run the two launches: fp32, K % 32 != 0
    fc = nn.Linear(48, 64).to(dev)
    layer = convnet.ConvLayer(fc, 'conv', (1, 1), plain=True)
    assert not convnet.fc_ln_fusable(layer, torch.zeros(1, 1, 4, 48, device=dev, dtype=dt))
    fc = nn.Linear(64, 64).to(dev)
    layer = convnet.ConvLayer(fc, 'conv', (1, 1), plain=True)
    assert not convnet.fc_ln_fusable(layer, torch.zeros(1, 1, 4, 64, device=dev))
    bad = L.msmc_fc_add_ln_fwd(None, None, None, None, None, None, None, None, None, None, None, 4, 64, 64, 1e-5, 0.0, None, 0, None)
    assert bad != 0


def check_two_graphs_over_one_bank(dev):
    """Two forward passes of ONE bank recorded before either is back-propagated, then two separate backward passes (the early
    delivery of hip/convnet.py counts open nodes per bank: the second pass closes the count the first one left open): the
    accumulated parameter gradients equal the stock operators', and the bank recognises the case (no early delivery fires while
    nodes of another graph are open) -- the round-4 advisor item on ``_open_nodes``"""
    import torch.nn as nn
    from msmctts_amd.hip import convnet
    from msmctts_amd.networks.layers import WNConv1d
    torch.manual_seed(5)
    conv_a, conv_b = WNConv1d(16, 16, 3, padding=1).to(dev), WNConv1d(16, 16, 5, padding=2).to(dev)
    la, lb = conv_a.hip_layer(), conv_b.hip_layer()
    bank = convnet.ConvBank([la, lb])
    x1 = torch.randn(2, 1, 50, 16, device=dev)
    x2 = torch.randn(2, 1, 50, 16, device=dev)
    fired = []
    orig = convnet.ConvBank._finish_backward

    def spy(self, early=False):
        fired.append(early)
        return orig(self, early=early)
    convnet.ConvBank._finish_backward = spy
    try:
        bank.prepare(torch.float32)
        y1 = convnet.hip_conv(bank, lb, convnet.hip_conv(bank, la, x1))
        y2 = convnet.hip_conv(bank, lb, convnet.hip_conv(bank, la, x2))
        y1.sum().backward()
        assert fired == [False], fired              # nodes of the second graph are still open: end-of-pass delivery only
        (2.0 * y2).sum().backward()
        assert True not in fired[:-1]
    finally:
        convnet.ConvBank._finish_backward = orig
    params = [('a.' + n, p) for n, p in conv_a.named_parameters()] + [('b.' + n, p) for n, p in conv_b.named_parameters()]
    mine = {n: p.grad.clone() for n, p in params}
    for m in (conv_a, conv_b):
        m.zero_grad()
    ref = lambda x: conv_b(conv_a(x.squeeze(1).transpose(1, 2)))
    (ref(x1).sum() + 2.0 * ref(x2).sum()).backward()
    for n, p in params:
        close(mine[n], p.grad, 2e-4 * max(1.0, float(p.grad.abs().max())), what='two graphs over one bank: ' + n)


def check_glue_kernels(dev):
    """csrc/norm.hip sum_n / dropout_add / row_mask against the stock operators they replace, and the grouped convolution's
    one-launch sum of the input gradients of members that share their input (hip/convnet.py SUM_SHARED_INPUTS)"""
    import torch.nn as nn
    from msmctts_amd.hip import convnet, norm
    from msmctts_amd.networks.layers import WNConv1d
    from msmctts_amd.utils.utils import get_mask_from_lengths
    torch.manual_seed(9)
    for dt, tol in ((torch.float32, 1e-6), (torch.bfloat16, 1.6e-2)):
        ts = [torch.randn(3, 1, 50, 16, device=dev).to(dt) for _ in range(4)]
        for n in (2, 3, 4):
            want = sum(t.float() for t in ts[:n])
            close(norm.sum_n(ts[:n]).float(), want, tol * max(1.0, float(want.abs().max())), what='sum_n %d' % n)
        odd = [torch.randn(7, device=dev).to(dt) for _ in range(3)]                  # (a size the kernel does not take: stock adds)
        close(norm.sum_n(odd).float(), sum(t.float() for t in odd), 4 * tol)
        # dropout + residual add
        x = torch.randn(4, 30, 64, device=dev).to(dt).requires_grad_(True)
        r = torch.randn(4, 30, 64, device=dev).to(dt).requires_grad_(True)
        y = norm.dropout_add(x, r, 0.0, norm.new_salt())
        close(y.float(), x.detach().float() + r.detach().float(), tol * 4)
        salt = norm.new_salt()
        y2 = norm.dropout_add(x.detach(), None, 0.25, salt)                            # the dropout alone: its zeros are the mask
        kept = (y2 != 0) | (x.detach() == 0)
        rate = float(kept.float().mean())
        assert abs(rate - 0.75) < 0.03, rate
        close(y2.float()[kept], x.detach().float()[kept] / 0.75, tol * 4 * max(1.0, float(x.detach().abs().max())))
        y = norm.dropout_add(x, r, 0.25, salt)                                         # same seed word, same salt: same mask
        close(y.float(), y2.float() + r.detach().float(), tol * 4 * max(1.0, float(y.detach().abs().max())))
        go = torch.randn_like(y)
        (y.float() * go.float()).sum().backward()
        live = (y2 != 0) & (go != 0)
        assert torch.equal((x.grad != 0) & (x.detach() != 0), live)                    # backward regenerated the forward's mask
        close(r.grad.float(), go.float(), 0.0)
        scale = (x.grad.float()[live] / go.float()[live])
        assert float((scale - 1.0 / 0.75).abs().max()) < 2e-2
        # row mask
        lens = torch.tensor([5, 0, 17, 9], device=dev, dtype=torch.int32)
        for L in (lens, lens.long()):
            keep = norm.row_mask(L, 17, dt)
            want = (~get_mask_from_lengths(L, 17)).to(dt)
            assert torch.equal(keep, want)
    # members of a grouped call that share their input: one summed gradient
    convs = [WNConv1d(16, 16, k, padding=k // 2).to(dev) for k in (3, 5, 7)]
    layers = [c.hip_layer() for c in convs]
    bank = convnet.ConvBank(layers)
    x0 = torch.randn(2, 1, 40, 16, device=dev)
    got = {}
    for flag in (True, False):
        saved, convnet.SUM_SHARED_INPUTS = convnet.SUM_SHARED_INPUTS, flag
        try:
            for c in convs:
                c.zero_grad()
            bank.prepare(torch.float32)
            x = x0.clone().requires_grad_(True)
            outs = convnet.hip_conv_group(bank, [dict(layer=l, x=x, in_slope=0.1) for l in layers])
            sum((o * (i + 1)).sum() for i, o in enumerate(outs)).backward()
            got[flag] = (x.grad.clone(), [p.grad.clone() for c in convs for p in c.parameters()])
        finally:
            convnet.SUM_SHARED_INPUTS = saved
    close(got[True][0], got[False][0], 2e-5 * max(1.0, float(got[False][0].abs().max())), what='shared-input gradient sum')
    for a, b in zip(got[True][1], got[False][1]):
        close(a, b, 1e-6 * max(1.0, float(b.abs().max())))
    xr = x0.clone().requires_grad_(True)
    act = torch.nn.functional.leaky_relu(xr.squeeze(1).transpose(1, 2), 0.1)
    sum((c(act) * (i + 1)).sum() for i, c in enumerate(convs)).backward()
    close(got[True][0], xr.grad, 2e-4 * max(1.0, float(xr.grad.abs().max())), what='shared-input gradient against stock operators')


def check_direct_kernels_with_every_epilogue_operand(dev):
    """conv_direct_small / outer / dot (descriptor variant 8) with bias, mask (data gradient), both residuals, the division and the
    output leaky-ReLU at once -- the vector run epilogue of round 6 (dir_epilogue_run) -- against the tiled second-generation kernel
    (variant 2) on the same operands, fp32 and bf16, channel counts that take the vector path (Cout = 4, 8, 16) and one that does not"""
    from msmctts_amd.hip import conv
    torch.manual_seed(21)
    B, H, W = 2, 9, 37
    saved = conv._tune
    state = {'variant': 8}
    ran = 0

    def forced(kind, desc, launch, candidates):
        desc.variant, desc.split_shift, desc._tuned = state['variant'], 0, True
    conv._tune = forced
    try:
        for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 1.5e-2)):
            for ci, co in ((2, 4), (4, 8), (1, 16), (8, 4), (2, 3), (64, 1), (1, 64)):
                x = torch.randn(B, H, W, ci, device=dev).to(dt)
                w = (torch.randn(9, co, ci, device=dev) / (9 * ci) ** 0.5).to(dt)
                bias = torch.randn(co, device=dev)
                res, res2 = torch.randn(B, H, W, co, device=dev).to(dt), torch.randn(B, H, W, co, device=dev).to(dt)
                geom = conv.Geometry(H, W, (3, 3), (1, 1), (1, 1), (1, 1), False)
                outs = []
                for v in (8, 2):
                    state['variant'] = v
                    conv._PLANS.clear()
                    geom.plans.clear()
                    try:
                        outs.append(conv.conv_forward(x, w, geom, bias=bias, in_slope=0.2, res=res, res2=res2, out_div=3.0, out_slope=0.1))
                    except RuntimeError:
                        outs.append(None)                       # (a shape the direct kernels do not take)
                if outs[0] is None:
                    continue
                ran += 1
                scale = max(1.0, float(outs[1].float().abs().max()))
                close(outs[0].float(), outs[1].float(), tol * scale, what='direct forward %d -> %d %s' % (ci, co, dt))
                # data gradient with the activation mask and a tap gradient as residual
                g = torch.randn(B, H, W, co, device=dev).to(dt)
                wb = w.transpose(1, 2).contiguous()
                mask, tapg = torch.randn(B, H, W, ci, device=dev).to(dt), torch.randn(B, H, W, ci, device=dev).to(dt)
                outs = []
                for v in (8, 2):
                    state['variant'] = v
                    conv._PLANS.clear()
                    geom.plans.clear()
                    try:
                        outs.append(conv.conv_dgrad(g, wb, geom, mask_src=mask, mask_slope=0.2, res=tapg, out_div=3.0))
                    except RuntimeError:
                        outs.append(None)
                if outs[0] is None:
                    continue
                ran += 1
                scale = max(1.0, float(outs[1].float().abs().max()))
                close(outs[0].float(), outs[1].float(), tol * scale, what='direct data gradient %d <- %d %s' % (ci, co, dt))
    finally:
        conv._tune = saved
        conv._PLANS.clear()
    assert ran >= 16, ran                                   # (the direct kernels took the cases: small, dot and outer, both dtypes)


def check_hip_adamw(dev):
    """csrc/optim.hip (grad-norm clip + AdamW of all tensors in three launches) against clip_grad_norm_ + torch.optim.AdamW
    over several steps, odd sizes and unaligned views; state_dict round trip both ways"""
    from msmctts_amd.trainers.optimizers.hip_adamw import HipAdamW
    torch.manual_seed(0)
    shapes = [(7,), (33, 5), (4096,), (5000,), (3, 3, 3)]
    base = torch.randn(20000, device=dev)
    mine = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    a = HipAdamW(mine, lr=2e-3, betas=(0.8, 0.99), eps=1e-8, weight_decay=0.01)
    b = torch.optim.AdamW(ref, lr=2e-3, betas=(0.8, 0.99), eps=1e-8, weight_decay=0.01)
    for step in range(4):
        for p, q in zip(mine, ref):
            g = torch.randn_like(p) * (3.0 if step % 2 == 0 else 0.01)
            p.grad = g.clone() if step != 2 else base[1:1 + g.numel()].view_as(g).clone()
            q.grad = p.grad.clone()
        norm = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        b.step()
        a.step(max_norm=1.0)
        assert abs(float(a.grad_norm) - float(norm)) <= 1e-5 * max(1.0, float(norm))
        for p, q in zip(mine, ref):
            assert (p - q).abs().max() <= 2e-6, step
            assert (p.grad - q.grad).abs().max() <= 1e-6 * max(1.0, float(q.grad.abs().max()))       # clipped in place
    sd = a.state_dict()
    assert set(sd['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'} and float(sd['state'][0]['step']) == 4.0
    b2 = torch.optim.AdamW([torch.nn.Parameter(p.detach().clone()) for p in mine], lr=2e-3, betas=(0.8, 0.99), weight_decay=0.01)
    b2.load_state_dict(sd)                                  # our checkpoint into torch's optimizer
    a2 = HipAdamW([torch.nn.Parameter(p.detach().clone()) for p in mine], lr=2e-3, betas=(0.8, 0.99), weight_decay=0.01)
    a2.load_state_dict(b.state_dict())                      # torch's checkpoint into ours
    for opt in (a2, b2):
        for p in opt.param_groups[0]['params']:
            p.grad = torch.ones_like(p) * 0.1
    a2.step()
    b2.step()
    for p, q in zip(a2.param_groups[0]['params'], b2.param_groups[0]['params']):
        assert (p - q).abs().max() <= 2e-6


def check_hip_adamw_resume_then_capture_rollback(dev):
    """a checkpoint loaded into HipAdamW and then put through the trainer's capture protocol (record the state tensors,
    run warm-up steps, copy the recorded values back) must come out with the LOADED step count and moments: the flat
    state is re-homed at load time, so the recorded tensors are the live ones (round-2 advisor finding: they were orphans,
    and the roll-back zeroed the live moments)"""
    from msmctts_amd.trainers.optimizers.hip_adamw import HipAdamW
    torch.manual_seed(1)
    shapes = [(9,), (17, 3), (4100,)]
    ps = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
    a = HipAdamW(ps, lr=1e-3, betas=(0.8, 0.99))
    for _ in range(3):
        for p in ps:
            p.grad = torch.randn_like(p)
        a.step(max_norm=1.0)
    sd = a.state_dict()
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    b = HipAdamW(qs, lr=1e-3, betas=(0.8, 0.99))
    b.load_state_dict(sd)
    recorded = [(v, v.detach().clone()) for st in b.state.values() for v in st.values() if torch.is_tensor(v)]
    assert recorded
    for _ in range(2):                                        # the capture warm-up really trains ...
        for q in qs:
            q.grad = torch.randn_like(q)
        b.step(max_norm=1.0)
    live = set(id(v) for st in b.state.values() for v in st.values() if torch.is_tensor(v))
    assert live == set(id(v) for v, _ in recorded), 'state tensors were replaced after load_state_dict'
    with torch.no_grad():
        for v, saved in recorded:                             # ... and is rolled back
            v.copy_(saved)
    for i, q in enumerate(qs):
        assert float(b.state[q]['step']) == 3.0
        assert torch.equal(b.state[q]['exp_avg'], sd['state'][i]['exp_avg'].to(dev))
        assert torch.equal(b.state[q]['exp_avg_sq'], sd['state'][i]['exp_avg_sq'].to(dev))
    # a child without trainable parameters is a no-op, not an IndexError
    frozen = [torch.nn.Parameter(torch.randn(5, device=dev), requires_grad=False)]
    c = HipAdamW(frozen, lr=1e-3)
    c.prepare()
    c.step()


def check_predictor_dropout_masks_advance(device):
    """PredictorTrainer.train_step advances the seed word of the fused dropout kernels once per step (the frozen
    autoencoder's analysis() does not), so consecutive steps draw different masks; with the word held fixed the same
    call reproduces its masks (what backward relies on)"""
    from msmctts_amd.hip import norm as hipnorm
    from msmctts_amd.tasks import build_task
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.trainers.optimizers import build_optimizer
    from msmctts_amd.utils.config import Config
    z = load_npz('small_predictor.npz')
    pc = small_predictor_cfg()
    for part in ('encoder_config', 'decoder_config', 'adaptor_config'):
        pc[part] = dict(pc[part], dropout=0.5, fused_layernorm=True)
    cfg = Config({'id': 'small_predictor_dropout', 'task': {'_name': 'MSMCTTS', '_mode': 'train_predictor', 'predictor': pc},
                  'trainer': dict(PREDICTOR_TRAINER, _name='PredictorTrainer'),
                  'optimizer': {'_default': dict(_name='Adam', learning_rate=0.0, betas=[0.9, 0.98], eps=1e-9, weight_decay=0)},
                  'dataset': dict(samplerate=24000, feature=['mel', 'wav'], frameshift=[300, 1])})
    task = build_task(cfg, mode='train').to(device).train()
    _, atask = build_small(device)
    batch = {k[len('batch.'):]: t(v).to(device) for k, v in z.items() if k.startswith('batch.')}
    tr = build_trainer(cfg, task, num_gpus=0, rank=0)
    tr.autoencoder = atask.autoencoder
    tr.optimizer = build_optimizer(task, cfg.optimizer)
    dev = batch['mel'].device
    w0 = int(hipnorm.seed_word(dev))
    losses = []
    for i in range(3):                                        # learning rate 0: only the masks differ between steps
        log = tr.train_step({k: v.clone() for k, v in batch.items()}, i)
        losses.append(float(log['loss']['total_loss']))
    assert int(hipnorm.seed_word(dev)) == w0 + 3
    assert len(set(losses)) == 3, ('identical dropout masks on consecutive predictor steps', losses)


def check_emb_autoencoder(device):
    """MSMCVQGANEmb (the QS-TTS synthesiser over speech-embedding frames, with the pitch / energy side encoder) against the
    reference's own module (tests/golden/small_emb.npz): state_dict surface, training-mode forward over windows (every
    entry of the output dictionary 1e-3, VQ indices exact, the input gradient of a scalar of the outputs), the codebooks
    after the EMA step, training-mode analysis, evaluation-mode analysis -> synthesis (both call forms) and window='full'."""
    from msmctts_amd.networks import find_modules
    z = load_npz('small_emb.npz')
    cfg = json_field(z['cfg'])
    (_, m), = find_modules({'autoencoder': dict(cfg, _name='MSMCVQGANEmb')})
    want_keys = [k[len('state.'):] for k in z if k.startswith('state.')]
    assert list(m.state_dict().keys()) == want_keys
    m.load_state_dict({k: t(z['state.' + k]) for k in want_keys})
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m = m.to(device).train()
    b = {k[len('batch.'):]: t(v).to(device) for k, v in z.items() if k.startswith('batch.')}
    windows = [tuple(int(v) for v in row) for row in z['windows']]

    def compare(prefix, d, skip=()):
        seen = 0
        for k, v in d.items():
            if k in skip:
                continue
            if torch.is_tensor(v):
                items = [('%s.%s' % (prefix, k), v)]
            elif isinstance(v, (tuple, list)):
                items = [('%s.%s.%d' % (prefix, k, i), x) for i, x in enumerate(v) if torch.is_tensor(x)]
            elif isinstance(v, dict):
                seen += compare('%s.%s' % (prefix, k), v)
                continue
            else:
                continue
            for name, got in items:
                want = z[name]
                if 'indices' in name or 'lengths' in name:
                    assert np.array_equal(got.cpu().numpy(), want), name
                else:
                    close(got, want, what=name)
                seen += 1
        return seen

    e = b['emb'].clone().requires_grad_(True)
    o = m(e, b['emb_length'], b['pitch'], b['energy'], window=windows)
    assert compare('train', o) >= 13
    scalar = (o['decoder_outputs'].pow(2).mean() + o['mel_outputs'].mean() + sum(d.mean() for d in o['encoder_diffs'])
              + o['decoder_diffs']['total_loss'] + o['content_representations'].mean())
    scalar.backward()
    close(scalar, z['train.scalar'], what='scalar')
    close(e.grad, z['train.grad_emb'], 1e-3 * max(1e-6, float(np.abs(z['train.grad_emb']).max())) + 1e-7, what='grad emb')
    sd = m.state_dict()
    for k in z:
        if k.startswith('after.'):
            ref = z[k]
            close(sd[k[len('after.'):]], ref, 1e-5 * max(1.0, float(np.abs(ref).max())), 1e-5, what=k)
    a = m.analysis(b['emb'], b['emb_length'], b['pitch'], b['energy'])
    assert compare('train_analysis', a, skip=('quantizer_states',)) >= 8 and 'quantizer_states' in a
    m.eval()
    with torch.no_grad():
        qs = m.analysis(b['emb'], b['emb_length'], b['pitch'], b['energy'])
        assert compare('eval_analysis', qs) >= 8
        close(m.synthesis(qs, qs['quantizer_lengths']), z['eval.wav'], what='synthesis(dict)')
        close(m.synthesis(list(qs['quantizer_outputs']), qs['quantizer_lengths']), z['eval.wav_from_sequences'],
              what='synthesis(sequences)')
        close(m(b['emb'], b['emb_length'], b['pitch'], b['energy'])['decoder_outputs'], z['eval.full.decoder_outputs'],
              what="window='full'")
    with pytest_raises(NotImplementedError):
        find_modules({'autoencoder': dict(cfg, _name='MSMCVQGANEmb', global_encoder_config={'_name': 'ECAPA_TDNN'})})


def pytest_raises(exc):
    import pytest
    return pytest.raises(exc)


def check_inference(device):
    """the task's inference glue (analysis-synthesis; text -> predictor -> synthesis with teacher durations) in evaluation
    mode against the reference's MSMCTTS.infer_step (tests/golden/small_infer.npz)"""
    from msmctts_amd.tasks import build_task
    from msmctts_amd.utils.config import Config
    z, zp, zb = load_npz('small_infer.npz'), load_npz('small_predictor.npz'), load_npz('small_modules.npz')

    def check(tag, wavs):
        for i, w in enumerate(wavs):
            want = z['%s.wav.%d' % (tag, i)]
            w = w.detach().double().reshape(-1).cpu()
            assert w.numel() == int(want[0]), (tag, i, w.numel(), want[0])
            assert abs(w.mean().item() - want[1]) <= TOL and abs(w.abs().mean().item() - want[2]) <= TOL
            close(w[:3000:3], want[3:], what='%s wav %d' % (tag, i))

    _, atask = build_small(device)
    atask.eval()
    mel, ml = t(zb['batch.mel']).to(device), t(zb['batch.mel_length']).to(device)
    with torch.no_grad():
        res = atask.infer_step({'mel': mel, 'mel_length': ml}, mode='train_autoencoder')
    check('ae', res['wav'])
    cfg = Config({'id': 'small_infer', 'task': {'_name': 'MSMCTTS', '_mode': 'train_predictor', 'predictor': small_predictor_cfg()},
                  'dataset': dict(samplerate=24000, feature=['mel', 'wav'], frameshift=[300, 1])})
    task = build_task(cfg, mode='infer')
    task.load_state_dict({k[len('state.'):]: t(v) for k, v in zp.items() if k.startswith('state.')})
    task = task.to(device).eval()
    task.autoencoder, task.load_modules = atask.autoencoder, True
    feed = {k: t(zp['batch.' + k]).to(device) for k in ('text', 'text_length', 'dur')}
    with torch.no_grad():
        res = task(feed)                                # mode 'infer' -> infer_step -> predict
    check('tts', res['wav'])
    close(res['embedding'], z['tts.embedding'], what='embedding')
    assert np.array_equal(res['duration'].cpu().numpy(), z['tts.duration'])


def check_attention(dev):
    """csrc/attn.hip (bf16, head size 64) against the reference's chain -- softmax(q k^T / sqrt(d) + key-padding mask),
    dropout, times v -- in fp32: output and the gradient of the fused projection (dq | dk | dv), ragged lengths, key tiles
    that are not multiples of 32, and with dropout: the kernel's own mask is recovered by a probe call (uniform
    probabilities, one-hot values) and the backward pass must have regenerated exactly that mask"""
    from msmctts_amd.hip import attn, norm
    torch.manual_seed(0)
    for (B, T, H, pd) in ((2, 45, 2, 0.0), (1, 64, 1, 0.0), (3, 33, 2, 0.0), (2, 50, 2, 0.25)):
        qkv = (torch.randn(B, T, H * 192, device=dev) * 0.7).bfloat16().requires_grad_(True)
        pos = torch.arange(1, T + 1, device=dev).repeat(B, 1)
        if B > 1:
            pos[1, T - 13:] = 0
        bias = attn.pad_key_bias(pos)
        assert bias.shape[1] % 32 == 0 and bool(torch.isinf(bias[:, T:]).all())
        salt = norm.new_salt()
        out = attn.attention(qkv, bias, H, 0.125, pd, salt)
        go = torch.randn(B, T, H * 64, device=dev)
        (out.float() * go).sum().backward()
        mask = torch.ones(B, H, T, T, device=dev)
        if pd > 0:
            probe = torch.zeros(B, T, H, 192, device=dev)
            for k in range(T):
                probe[:, k, :, 128 + k] = 1.0
            flat = torch.zeros_like(bias)
            flat[:, T:] = float('-inf')
            o = attn.attention(probe.reshape(B, T, H * 192).bfloat16(), flat, H, 0.0, pd, salt)
            mask = (o.float().reshape(B, T, H, 64)[..., :T].permute(0, 2, 1, 3) > 0).float()
            assert abs(mask.mean().item() - (1 - pd)) < 0.03
        x = qkv.detach().float().reshape(B, T, H, 192).requires_grad_(True)
        s = torch.einsum('bqhd,bkhd->bhqk', x[..., :64], x[..., 64:128]) * 0.125 + bias[:, None, None, :T]
        ref = torch.einsum('bhqk,bkhd->bqhd', torch.softmax(s, -1) * mask / (1 - pd), x[..., 128:]).reshape(B, T, H * 64)
        (ref * go).sum().backward()
        close(out, ref, 2e-2, what='attention out')
        g, gr = qkv.grad.float().reshape(B, T, H, 192), x.grad
        for name, a, b in (('dq', 0, 64), ('dk', 64, 128), ('dv', 128, 192)):
            close(g[..., a:b], gr[..., a:b], 2e-2 * max(1.0, float(gr[..., a:b].abs().max())), what=name)
    # edge cases: a single frame, exactly one / one-plus-one key tile, four heads, an utterance with ONE valid key
    for (B, T, H) in ((1, 1, 1), (2, 32, 4), (2, 33, 1), (2, 129, 2)):
        qkv = torch.randn(B, T, H * 192, device=dev).bfloat16().requires_grad_(True)
        pos = torch.arange(1, T + 1, device=dev).repeat(B, 1)
        if B > 1:
            pos[1, 1:] = 0
        bias = attn.pad_key_bias(pos)
        out = attn.attention(qkv, bias, H, 0.125)
        out.float().sum().backward()
        assert bool(torch.isfinite(out.float()).all()) and bool(torch.isfinite(qkv.grad.float()).all()), (B, T, H)
        x = qkv.detach().float().reshape(B, T, H, 192)
        s = torch.einsum('bqhd,bkhd->bhqk', x[..., :64], x[..., 64:128]) * 0.125 + bias[:, None, None, :T]
        ref = torch.einsum('bhqk,bkhd->bqhd', torch.softmax(s, -1), x[..., 128:]).reshape(B, T, H * 64)
        close(out, ref, 2e-2, what='attention edge %s' % ((B, T, H),))
        if B > 1:               # every query of the one-key utterance returns that key's value
            close(out[1].float().reshape(T, H, 64), x[1, :1, :, 128:].expand(T, H, 64), 1e-2, what='single key')


def check_resblock_standalone(dev):
    """ResBlock1 called on its own (reference hifigan/common.py:44-51) against the stock operator chain: output, input
    gradient and every parameter gradient"""
    import torch.nn.functional as F
    from msmctts_amd.networks.hifigan.common import ResBlock1
    torch.manual_seed(5)
    for C, k, L in ((32, 3, 77), (16, 7, 40)):
        rb = ResBlock1(C, k, (1, 3, 5)).to(dev)
        x = torch.randn(2, C, L, device=dev, requires_grad=True)
        y = rb(x)
        go = torch.randn_like(y)
        (y * go).sum().backward()
        got = {n: p.grad.clone() for n, p in rb.named_parameters()}
        gx = x.grad.clone()
        rb.zero_grad()
        xr = x.detach().clone().requires_grad_(True)
        h = xr
        for c1, c2 in zip(rb.convs1, rb.convs2):
            t = F.conv1d(F.leaky_relu(h, 0.1), c1.weight(), c1.bias, 1, c1.padding, c1.dilation)
            h = F.conv1d(F.leaky_relu(t, 0.1), c2.weight(), c2.bias, 1, c2.padding, c2.dilation) + h
        (h * go).sum().backward()
        close(y, h, 2e-4, what='resblock out')
        close(gx, xr.grad, 2e-4, 1e-3, what='resblock gx')
        for n, p in rb.named_parameters():
            close(got[n], p.grad, 2e-4, 2e-3, what=n)


def check_weight_image_tiles(dev):
    """msmc_wn_prepare_multi_tiles (norms-only row pass + one tiled pass writing both kernel layouts) against the definition
    w = v * g / ||v|| (plain layers: w = v) over shapes that end inside a tile on either axis, every tap count the tile rule
    distinguishes (1, 2, 3-4, 5+) up to the limit, convolution and transposed-convolution stride sets, fp32 and bf16 images,
    several items in one call -- and against the previous two-pass form msmc_wn_prepare_multi_tiled on the same items."""
    import ctypes
    from msmctts_amd.hip import lib
    L = lib.get()
    torch.manual_seed(23)
    shapes = [(70, 33, 1, True, 'conv'), (64, 128, 1, False, 'conv'), (5, 7, 2, True, 'convT'), (96, 40, 3, False, 'conv'),
              (33, 65, 4, True, 'conv'), (40, 31, 5, True, 'convT'), (17, 50, 7, False, 'conv'), (32, 32, 9, True, 'conv'),
              (130, 20, 11, True, 'conv'), (9, 70, 12, True, 'convT'), (3, 3, 16, False, 'conv'), (1, 1, 1, True, 'conv')]
    for dtype, tol in ((torch.float32, 3e-7), (torch.bfloat16, 4e-3)):
        esz = 4 if dtype == torch.float32 else 2
        vs = [torch.randn(A, Bc, T, device=dev) for A, Bc, T, _, _ in shapes]
        gs = [torch.rand(A, device=dev) + 0.5 for A, _, _, _, _ in shapes]
        pad8 = lambda n: (n + 7) // 8 * 8
        tot = sum(pad8(v.numel()) for v in vs)

        def run(entry):
            w1 = torch.full((tot,), float('nan'), dtype=dtype, device=dev)
            w2 = torch.full((tot,), float('nan'), dtype=dtype, device=dev)
            inv = torch.zeros(sum(v.shape[0] for v in vs), device=dev)
            items = (lib.WnItem * len(shapes))()
            ow = oa = blk = tblk = 0
            for it, v, g, (A, Bc, T, normed, kind) in zip(items, vs, gs, shapes):
                it.v, it.g = v.data_ptr(), (g.data_ptr() if normed else None)
                it.dst1, it.dst2 = w1.data_ptr() + ow * esz, w2.data_ptr() + ow * esz
                it.inv_norm = inv.data_ptr() + oa * 4
                it.A, it.Bc, it.T, it.dtype = A, Bc, T, (0 if dtype == torch.float32 else 1)
                it.block0, it.tblock0 = blk, tblk
                # conv: v (Cout, Cin, T) -> [T][Cout][Cin] and [T][Cin][Cout]; convT: v (Cin, Cout, T) -> [T][Cin][Cout] and [T][Cout][Cin]
                it.s1[0], it.s1[1], it.s1[2] = A * Bc, Bc, 1
                it.s2[0], it.s2[1], it.s2[2] = A * Bc, 1, A
                blk += A
                tblk += (int(L.msmc_wn_tile_blocks(A, Bc, T)) if entry == 'tiles' else ((A + 63) // 64) * ((Bc + 15) // 16))
                ow, oa = ow + pad8(v.numel()), oa + A
            dev_items = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(dev)
            st = lib.stream(w1)
            if entry == 'tiles':
                from msmctts_amd.hip import convnet
                convnet._wn_prepare(dev_items, convnet._wn_maps(items, torch.device(dev)), len(shapes), blk, tblk,
                                    max(s[2] for s in shapes), st, 'msmc_wn_prepare_multi_tiles')
            else:
                lib.check(L.msmc_wn_prepare_multi_tiled(lib.ptr(dev_items), len(shapes), blk, tblk, st), 'msmc_wn_prepare_multi_tiled')
            return w1, w2, inv
        w1, w2, inv = run('tiles')
        o1, o2, oinv = run('tiled')
        ow = oa = 0
        for v, g, (A, Bc, T, normed, kind) in zip(vs, gs, shapes):
            n = v.numel()
            norm = v.reshape(A, -1).double().norm(dim=1)
            w = (v.double() * (g.double() / norm).view(A, 1, 1)) if normed else v.double()
            want1 = w.permute(2, 0, 1).reshape(-1)                   # [T][A][Bc]
            want2 = w.permute(2, 1, 0).reshape(-1)                   # [T][Bc][A]
            what = '%s %dx%dx%d %s' % (str(dtype).split('.')[-1], A, Bc, T, 'wn' if normed else 'plain')
            close(w1[ow:ow + n].double(), want1, tol, tol, 'layout 1 ' + what)
            close(w2[ow:ow + n].double(), want2, tol, tol, 'layout 2 ' + what)
            close(o1[ow:ow + n].double(), want1, tol, tol, 'previous form, layout 1 ' + what)
            close(o2[ow:ow + n].double(), want2, tol, tol, 'previous form, layout 2 ' + what)
            if normed:
                close(inv[oa:oa + A].double(), 1.0 / norm, 3e-7, 3e-7, 'inv_norm ' + what)
            assert not torch.isnan(w1[ow:ow + n].float()).any() and not torch.isnan(w2[ow:ow + n].float()).any(), what
            pad = pad8(n) - n
            if pad:                                                   # nothing written past an item's image
                assert torch.isnan(w1[ow + n:ow + n + pad].float()).all() and torch.isnan(w2[ow + n:ow + n + pad].float()).all(), what
            ow, oa = ow + pad8(n), oa + A


def check_window_gather_and_output_tanh(dev):
    """msmc_window_gather against the operator chain it replaces (VQGANTrainer._build_windows: arange + add, multiply + arange +
    add, gather) and msmc_tanh_f32_fwd / _bwd against ``tanh(x.float())`` and its gradient, fp32 and bf16 inputs"""
    from msmctts_amd.hip import norm as hipnorm
    from msmctts_amd.hip import spectral
    g = torch.Generator().manual_seed(9)
    B, L, fl, hop = 5, 4000, 7, 300
    wav = torch.randn(B, L, generator=g).to(dev)
    starts = torch.tensor([0, 3, 6, 1, 2], dtype=torch.int64, device=dev)          # (6 + 7) * 300 = 3900 <= L
    frames, target = spectral.window_gather(starts, wav, fl, hop)
    want_f = starts.unsqueeze(1) + torch.arange(fl, device=dev).unsqueeze(0)
    sidx = (starts * hop).unsqueeze(1) + torch.arange(fl * hop, device=dev).unsqueeze(0)
    assert torch.equal(frames, want_f) and torch.equal(target, torch.gather(wav, 1, sidx))
    for dtype, tol in ((torch.float32, 1e-6), (torch.bfloat16, 1e-2)):
        x = torch.randn(3, 1, 501, 1, generator=g).to(dev).to(dtype).requires_grad_(True)
        go = torch.randn(3, 1, 501, 1, generator=g).to(dev)
        y = hipnorm.tanh_f32(x)
        assert y.dtype == torch.float32
        (y * go).sum().backward()
        xr = x.detach().clone().requires_grad_(True)
        yr = torch.tanh(xr.float())
        (yr * go).sum().backward()
        close(y, yr, 1e-6, what='tanh_f32 %s' % dtype)
        assert x.grad.dtype == dtype
        close(x.grad.float(), xr.grad.float(), tol, tol, what='tanh_f32 gradient %s' % dtype)


def check_fronts_lockstep(dev):
    """hip/spectral.py mrd_fronts / backward_rows_lockstep (every stage of several front-end chains in one launch: msmc_spectral_multi,
    grouped split-bf16 constant GEMMs) against the chains run one after the other (MrdFront / backward_rows): the same kernels'
    bodies on the same data -- bit-identical images, intermediates and waveform gradients, in the fp32 and the bf16 configuration,
    with and without a filter bank, on a row range"""
    from msmctts_amd.hip import spectral
    from msmctts_amd.utils.audio import TorchSTFT
    g = torch.Generator().manual_seed(21)
    B, L = 6, 2400
    x = torch.randn(B, L, generator=g).to(dev)
    stfts = [TorchSTFT(fft_size=h * 4, hop_size=h, win_size=h * 4, normalized=True, domain='double', mel_scale=(h != 30),
                       sample_rate=24000) for h in (15, 30, 50, 120)]
    for dtype in (torch.float32, torch.bfloat16):
        specs = [(s_.fft_size, s_.hop_size) + tuple(s_.consts(x.device)) for s_ in stfts]
        together = spectral.mrd_fronts(x, specs, dtype)
        alone = [spectral.MrdFront(x, n_fft, hop, dft, fb, dtype) for n_fft, hop, dft, fb in specs]
        for a, b in zip(together, alone):
            for name in ('spec', 'mag', 'mel', 'img'):
                assert torch.equal(getattr(a, name), getattr(b, name)), (dtype, a.hop, name)
        gs = [torch.randn(3, f.F, f.T, 2, generator=g).to(dev).to(dtype) for f in alone]
        gs[1] = None                                                    # (a front whose image nobody differentiated)
        got = spectral.backward_rows_lockstep(together, gs, 2, 5)
        for a, b, gi in zip(got, alone, gs):
            if gi is None:
                assert a is None
            else:
                assert torch.equal(a, b.backward_rows(gi, 2, 5)), (dtype, b.hop)


def check_codebook_split_update(dev):
    """msmc_vq_ema_stats + msmc_vq_ema_apply (the two halves around the cross-rank sum of sync_codebook_stats) are, on
    one rank, bit for bit the fused msmc_vq_ema_update"""
    from msmctts_amd.hip import vq as hipvq
    from msmctts_amd.networks.vqgantts.modules import MultiHeadQuantize
    g = torch.Generator().manual_seed(21)
    x = torch.randn(4, 137, 64, generator=g).to(dev)
    ln = torch.tensor([137, 20, 5, 131]).to(dev)
    torch.manual_seed(3)
    q1 = MultiHeadQuantize(64, 32, 4).to(dev).train()
    torch.manual_seed(3)
    q2 = MultiHeadQuantize(64, 32, 4).to(dev).train()
    q2.sync_stats = True
    for _ in range(3):
        q1(x, ln, update=True)
        q2(x, ln, update=True)
        assert len(hipvq.PENDING) == 1
        hipvq.flush_codebook_sync()
    for (k, a), b in zip(q1.state_dict().items(), q2.state_dict().values()):
        assert torch.equal(a, b), k


def check_mr_stft(device):
    """MultiResolutionSTFTLoss of the product against the reference values in frontends.npz."""
    from msmctts_amd.trainers.criterions.stft_loss import MultiResolutionSTFTLoss
    z = load_npz('frontends.npz')
    wav, wav2 = t(z['wav']).to(device), t(z['wav2']).to(device).requires_grad_(True)
    r = MultiResolutionSTFTLoss()(wav2, wav)
    for k, ref in (('sc_loss', 'mrstft.sc'), ('mag_loss', 'mrstft.mag')):
        got, want = float(r[k]), float(z[ref])
        assert abs(got - want) <= 1e-4 * max(1.0, abs(want)), (k, got, want)
    (r['sc_loss'] + r['mag_loss']).backward()           # gradient flows through the HIP front-end
    g = wav2.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    # same gradient as autograd through the oracle's torch.stft formulation
    from oracle import audio
    w2 = t(z['wav2']).clone().requires_grad_(True)
    ro = audio.mr_stft_loss(w2, t(z['wav']))
    (ro['sc_loss'] + ro['mag_loss']).backward()
    scale = float(w2.grad.abs().max())
    close(g.cpu(), w2.grad, 2e-3 * scale, 0.0, what='d(mr_stft)/d(wav) (scale %.2e)' % scale)


def check_vq_edge_cases(device):
    """Empty input, a single frame, utterances of length 0 (all padding) next to full ones, ragged lengths: the product
    quantiser against the oracle (same codebook, same data) -- indices exact, buffers after the EMA step equal."""
    from msmctts_amd.networks.vqgantts.modules import MultiHeadQuantize
    from oracle.vq import multi_head_quantize
    torch.manual_seed(11)
    H, K, D = 4, 64, 64
    for B, T, lens in ((3, 9, [9, 0, 4]), (1, 1, [1]), (2, 5, [0, 0]), (4, 33, [33, 17, 1, 32])):
        q = MultiHeadQuantize(D, K, H).train()
        heads = []
        for m in q.quantizers:
            e = torch.randn(D // H, K)
            m.embed.copy_(e)
            m.embed_avg.copy_(e)
            m.cluster_size.fill_(0.5)
            heads.append([e.clone(), torch.full((K,), 0.5), e.clone()])
        q = q.to(device)
        x = torch.randn(B, T, D)
        ln = torch.tensor(lens, dtype=torch.int64)
        qq, dd, ii = q(x.to(device), ln.to(device), update=True)
        q0, d0, i0 = multi_head_quantize(x, ln, heads, True)
        assert np.array_equal(ii.cpu().numpy(), i0.numpy()), (B, T, lens)
        close(qq, q0, 1e-5, what='quant')
        close(dd, d0, 1e-5, 1e-5, what='diff')
        for h, m in enumerate(q.quantizers):
            close(m.cluster_size, heads[h][1], 1e-6, 1e-6, what='cluster_size %s' % (lens,))
            close(m.embed_avg, heads[h][2], 1e-5, 1e-5, what='embed_avg %s' % (lens,))
            close(m.embed, heads[h][0], 1e-5, 1e-5, what='embed %s' % (lens,))
    # empty input: nothing to search, outputs keep their shapes, buffers untouched
    q = MultiHeadQuantize(D, K, H).to(device).train()
    before = [m.embed.clone() for m in q.quantizers]
    qq, dd, ii = q(torch.zeros(0, 7, D, device=device), torch.zeros(0, dtype=torch.int64, device=device), update=True)
    assert qq.shape == (0, 7, D) and dd.shape == (0, 7, D // H) and ii.shape == (0, 7, H)
    for m, b in zip(q.quantizers, before):
        assert torch.equal(m.embed, b)


def check_fft_prologue(dev, B=5, T=45, C=24):
    """csrc/norm.hip fft_prologue_kernel (MSMC_FFT_PROLOGUE=1, off by default): positions from lengths, positional-embedding
    add, cast, row mask and key-padding bias in one launch -- bit for bit the chain of stock operators it replaces
    (reference acoustic_models/transformer.py FFTBlocks.forward head, vqgantts/msmc_vqgan.py:56-58)"""
    from msmctts_amd.hip import attn as hipattn, norm as hipnorm
    from msmctts_amd.networks.acoustic_models.transformer import get_sinusoid_encoding_table
    torch.manual_seed(4)
    table = get_sinusoid_encoding_table(T + 3, C, padding_idx=0).to(dev)
    for len_dtype in (torch.int32, torch.int64):
        lengths = torch.tensor(([T, 1, 17, T - 1, 30] * B)[:B], dtype=len_dtype).to(dev)
        steps = torch.arange(1, T + 1, device=dev).unsqueeze(0)
        pos = steps * (steps <= lengths.unsqueeze(1))
        for in_dt, out_dt in ((torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16)):
            seq = torch.randn(B, T, C).to(dev, in_dt).requires_grad_(True)
            out, keep_row, bias = hipnorm.fft_prologue(seq, lengths, table, out_dt, (T + 31) // 32 * 32)
            want = (seq.detach() + table[pos]).to(out_dt)
            assert torch.equal(out, want), (in_dt, out_dt)
            assert torch.equal(keep_row.view(B, T), pos.ne(0).to(torch.uint8))
            assert torch.equal(bias, hipattn.pad_key_bias(pos))
            g = torch.randn(B, T, C).to(dev, out_dt)
            out.backward(g)
            assert seq.grad.dtype == in_dt and torch.equal(seq.grad, g.to(in_dt))
    seq = torch.randn(2, 7, 6).to(dev)                      # channel count without vector accesses, no bias
    out, keep_row = hipnorm.fft_prologue(seq, torch.tensor([7, 3]).to(dev), table[:, :6].contiguous(), torch.float32)
    pos = torch.tensor([[1, 2, 3, 4, 5, 6, 7], [1, 2, 3, 0, 0, 0, 0]]).to(dev)
    assert torch.equal(out, seq + table[:, :6][pos]) and torch.equal(keep_row.view(2, 7), pos.ne(0).to(torch.uint8))


def check_masked_mean_and_colsum(dev):
    """csrc/losses.hip masked_mean (QuantizerLoss / frame loss / 'mse' embedding loss) and the non-atomic column sums against
    the stock operator chains they replace: values, gradients, int32 / int64 lengths, bf16 operands, zero-length rows"""
    import torch.nn.functional as F
    from msmctts_amd.hip import conv as K
    from msmctts_amd.hip import losses
    from msmctts_amd.utils.utils import get_mask_from_lengths
    torch.manual_seed(2)
    for (B, T, C, ldt) in ((3, 24, 8, torch.int64), (5, 100, 64, torch.int32), (2, 7, 80, torch.int64), (16, 400, 80, torch.int64)):
        lengths = torch.randint(1, T + 1, (B,), device=dev).to(ldt)
        lengths[0] = T
        if B > 2:
            lengths[B - 1] = 0
        x = torch.randn(B, T, C, device=dev, requires_grad=True)
        y = torch.randn(B, T, C, device=dev)
        pad = get_mask_from_lengths(lengths, T).unsqueeze(-1)
        # mode 0: QuantizerLoss term
        want = x.masked_fill(pad, 0).sum() / lengths.sum() / C
        gw, = torch.autograd.grad(want * 3.0, x)
        got = losses.masked_mean(x, lengths)
        gg, = torch.autograd.grad(got * 3.0, x)
        close(got, want, 1e-5, what='masked mean')
        close(gg, gw, 1e-6, what='masked mean grad')
        # mode 1: frame loss (fp32 target, prediction in fp32 or bf16)
        for adt in (torch.float32, torch.bfloat16):
            a = x.detach().to(adt).requires_grad_(True)
            ml = F.mse_loss(y, a.float(), reduction='none').masked_fill(pad, 0)
            want = ml.sum() / lengths.sum() / C
            gw, = torch.autograd.grad(want, a)
            got = losses.masked_mean(a, lengths, b=y)
            gg, = torch.autograd.grad(got, a)
            close(got, want, 1e-5, what='masked mse')
            close(gg, gw, 1e-6 if adt == torch.float32 else 2e-3 * float(gw.float().abs().max()), what='masked mse grad')
            assert float(gg.float()[0, T - 1].abs().sum()) > 0 and (B <= 2 or float(gg.float()[B - 1].abs().sum()) == 0.0)
    for rows, C, dt in ((3840, 256, torch.float32), (19200, 128, torch.bfloat16), (5000, 64, torch.float32), (777, 32, torch.bfloat16),
                        (100, 24, torch.float32), (3, 512, torch.float32)):
        g = torch.randn(rows, C, device=dev).to(dt)
        want = g.float().sum(0)
        close(K.colsum(g), want, 1e-4 * max(1.0, float(want.abs().max())), what='colsum')
        acc = torch.ones(C, device=dev)
        K.colsum(g, out=acc)
        close(acc, want + 1.0, 1e-4 * max(1.0, float(want.abs().max())), what='colsum accumulate')
        assert torch.equal(K.colsum(g), K.colsum(g))                  # fixed summation order


def check_gan_loss_kernels(dev):
    """csrc/losses.hip multi-tensor L1 (feature matching) and MSE-to-constant (LSGAN) sums against the per-tensor stock
    operators of reference msmctts_trainer.py:165-171,187-193: values and gradients, fp32 / bf16, sizes that are not
    multiples of the 16-byte vectors, permuted (dense) views and operands at unaligned addresses (scalar path)"""
    import torch.nn.functional as F
    from msmctts_amd.hip import losses
    torch.manual_seed(4)
    for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1e-2)):
        shapes = [(3, 5, 7, 2), (2, 16, 33, 4), (4, 1, 1000), (1, 9), (2, 8, 64, 8)]
        fake = [torch.randn(s, device=dev).to(dt).requires_grad_(True) for s in shapes]
        real = [torch.randn(s, device=dev).to(dt) for s in shapes]
        fake_v = [fake[0].permute(0, 3, 1, 2), fake[1].permute(0, 3, 1, 2), fake[2], fake[3], fake[4].permute(0, 3, 1, 2)]
        real_v = [real[0].permute(0, 3, 1, 2), real[1].permute(0, 3, 1, 2), real[2], real[3], real[4].permute(0, 3, 1, 2)]
        got = losses.l1_sum(fake_v, real_v)
        want = sum(F.l1_loss(a.float(), b.float()) for a, b in zip(fake_v, real_v))
        close(got, want, tol, what='l1 sum')
        gg = torch.autograd.grad(got * 2.0, fake)
        gw = torch.autograd.grad(want * 2.0, fake)
        for a, b in zip(gg, gw):
            close(a, b, tol * max(1e-3, float(b.float().abs().max())) if dt == torch.bfloat16 else 1e-7, what='l1 grad')
        for target in (0.0, 1.0):
            got = losses.mse_const_sum(fake_v, target)
            want = sum(((a.float() - target) ** 2).mean() for a in fake_v)
            close(got, want, tol, what='mse const sum')
            gg = torch.autograd.grad(got, fake)
            gw = torch.autograd.grad(want, fake)
            for a, b in zip(gg, gw):
                close(a, b, tol * max(1e-3, float(b.float().abs().max())) if dt == torch.bfloat16 else 1e-6, what='mse grad')
        # operands at unaligned addresses: halves of a batch whose row size is not a multiple of 16 bytes
        base = torch.randn(4, 3, 5, device=dev).to(dt).requires_grad_(True)
        other = torch.randn(4, 3, 5, device=dev).to(dt)
        got = losses.l1_sum([base[1:3]], [other[1:3]])
        want = F.l1_loss(base[1:3].float(), other[1:3].float())
        close(got, want, tol, what='l1 sum (offset view)')
        (ga,), (gb,) = torch.autograd.grad(got, base), torch.autograd.grad(want, base)
        close(ga, gb, tol * max(1e-3, float(gb.float().abs().max())) if dt == torch.bfloat16 else 1e-7, what='l1 grad (offset view)')


def check_split_constant_gemm(dev):
    """csrc/gemm1.inc conv_gemm1s_kernel (variants 36 / 37: fp32 data times a pre-split constant matrix, three bf16
    products with fp32 accumulation) against the fp64 product: relative error of the two-piece split (~2^-16 of |w||x| per
    product), both tile shapes, ragged row counts, contractions that are not whole chunks, ring and all-in-flight paths; and
    the spectral chain built on it (image + gradient) against the exact-fp32 chain."""
    import ctypes
    from msmctts_amd.hip import conv as K
    from msmctts_amd.hip import lib, spectral
    torch.manual_seed(6)
    for (B, T, Cin, Cout) in ((2, 51, 960, 964), (3, 17, 60, 64), (1, 130, 484, 484), (2, 40, 1200, 2052), (1, 5, 128, 40)):
        x = torch.randn(B, 1, T, Cin, device=dev)
        w = torch.randn(1, Cout, Cin, device=dev) / Cin ** 0.5
        want = (x.double().reshape(-1, Cin) @ w[0].double().t()).reshape(B, 1, T, Cout)
        img = spectral.split_image(w)
        for variant in (36, 37):
            key = ('g1s', B, T, Cin, Cout)
            K._PLANS.pop(key, None)
            out = K.const_gemm_split(x, img, Cout)                      # (builds the descriptor)
            d = K._PLANS[key]
            d.variant, d._tuned = variant, True
            out = K.const_gemm_split(x, img, Cout)
            assert b'conv_gemm1s_kernel' in lib.get().msmc_conv_last_kernel()
            err = (out.double() - want).abs().max().item()
            scale = float((x.double().reshape(-1, Cin).abs() @ w[0].double().abs().t()).max())
            assert err <= 3.0e-5 * scale, (B, T, Cin, Cout, variant, err, scale)
    # the MRD image chain in split mode against exact fp32: values and waveform gradient
    n_fft, hop = 240, 60
    win = torch.hann_window(n_fft, device=dev)
    dft = spectral.dft_basis(n_fft, win, True, dev)
    F = n_fft // 2 + 1
    fb = spectral.projection(torch.rand(F, F).clamp(1e-6, 1.0) * (torch.rand(F, F) < 0.05), dev)
    wav = (torch.rand(3, 2400, device=dev) * 2 - 1)
    outs = []
    for split in (False, True):
        x = wav.clone().requires_grad_(True)
        front = spectral.MrdFront(x, n_fft, hop, dft, fb, torch.float32, split=split)
        g = torch.randn(front.img.shape, generator=torch.Generator().manual_seed(1)).to(dev)
        outs.append((front.img.clone(), front.backward_rows(g, 0, 3)))
    (img0, gx0), (img1, gx1) = outs
    close(img1, img0, 2e-4, what='split-bf16 MRD image')
    rel = ((gx1 - gx0).norm() / gx0.norm()).item()
    assert rel <= 2e-4, 'split-bf16 MRD waveform gradient: relative L2 error %.3e' % rel


def check_wave_fan(dev):
    """hip/spectral.py wave_fan (msmc_wave_fan_fwd / _bwd) against the stock chain it replaces in the discriminator: cast,
    F.pad(.., 'reflect') to a multiple of each period, and the autograd engine's sum of all consumers' gradients"""
    import torch.nn.functional as F
    from msmctts_amd.hip import spectral
    torch.manual_seed(8)
    for dtype, tol in ((torch.float32, 1e-6), (torch.bfloat16, 1e-2)):
        for (B, L, periods, n_alias) in ((3, 2400, (2, 3, 5, 7, 11), 5), (2, 37, (2, 3, 5), 0), (1, 100, (7,), 2)):
            y = (torch.rand(B, L, device=dev) * 2 - 1).requires_grad_(True)
            padded = [(L + p - 1) // p * p for p in periods]
            wavs, copies = spectral.wave_fan(y, n_alias, padded, dtype)
            yr = y.detach().clone().requires_grad_(True)
            ref = []
            for p, lp in zip(periods, padded):
                x = yr.unsqueeze(1).to(dtype)
                if lp != L:
                    x = F.pad(x, (0, lp - L), 'reflect')
                ref.append(x.squeeze(1))
            for a, b in zip(copies, ref):
                assert a.shape == b.shape and torch.equal(a, b), 'padded copy differs'
            for w in wavs:
                assert torch.equal(w, y)
            gen = torch.Generator().manual_seed(3)
            gc = [torch.randn(c.shape, generator=gen).to(dev).to(dtype) for c in copies]
            gw = [torch.randn(B, L, generator=gen).to(dev) for _ in wavs]
            loss = sum((c.float() * g.float()).sum() for c, g in zip(copies[:-1], gc[:-1])) + sum((w * g).sum() for w, g in zip(wavs, gw))
            lref = sum((c.float() * g.float()).sum() for c, g in zip(ref[:-1], gc[:-1])) + sum((yr * g).sum() for g in gw)
            loss.backward()                      # (the last copy has no consumer: its gradient arrives as None)
            lref.backward()
            close(y.grad, yr.grad, tol * max(1.0, float(yr.grad.abs().max())), what='wave fan gradient')


def check_reducer_stream_order(device, delay_cycles=int(2e8)):
    """A stock module FOLLOWED by a convolution bank in one reducer bucket, the way predictor-style models are built (stock
    ``nn.Linear`` between bank-backed stacks): in the backward pass the bank delivers its gradients EARLY from its side stream
    (hip/convnet.py FINISH_SIDE) -- held back here by ``delay_cycles`` on that stream -- and the stock module's gradients
    arrive afterwards on the calling stream and complete the bucket there.  Returns the largest deviation of the averaged
    gradients (one rank: the identity) from the same backward without a reducer.  Needs an initialised process group."""
    import torch.nn as nn
    from msmctts_amd.distributed.distributed import GradReducer
    from msmctts_amd.hip import convnet
    from msmctts_amd.networks.layers import WNConv1d

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.pre = nn.Linear(64, 64)                    # stock operator: its gradients are ready on the calling stream
            self.conv = WNConv1d(64, 64, 3, padding=1)      # convolution bank: gradients delivered by hand
            self._layer = self.conv.hip_layer()
            self._bank = convnet.ConvBank([self._layer])

        def forward(self, x):
            self._bank.prepare(torch.float32)
            return convnet.hip_conv(self._bank, self._layer, self.pre(x).unsqueeze(1).contiguous())

    torch.manual_seed(11)
    model = nn.ModuleDict({'net': Net()}).to(device)
    x = torch.randn(4, 200, 64, device=device)
    go = torch.randn(4, 1, 200, 64, device=device)

    def grads():
        model.zero_grad()
        (model['net'](x) * go).sum().backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    want = grads()                                          # no reducer, no delay
    assert convnet.FINISH_SIDE and convnet.EARLY_FINISH and convnet.STREAMS_ENABLED
    seen = []
    orig = convnet.ConvBank._finish_backward

    def slow(self, early=False):
        if early and self.w1.is_cuda:
            seen.append(torch.cuda.current_stream(self.w1.device))
            torch.cuda._sleep(delay_cycles)                 # the delivery's launches sit behind this on the side stream
        return orig(self, early=early)
    reducer = GradReducer(model, bucket_bytes=1 << 30)      # everything in ONE bucket
    keep_hook = convnet.GRAD_READY_HOOK
    convnet.GRAD_READY_HOOK = reducer._on_grad
    convnet.ConvBank._finish_backward = slow
    try:
        for p in model.parameters():                        # poison what a too-early concatenation would read
            if p.grad is not None:
                p.grad.fill_(float('nan'))
        model.zero_grad()
        (model['net'](x) * go).sum().backward()
        reducer.finish()
        torch.cuda.synchronize()
    finally:
        convnet.ConvBank._finish_backward = orig
        convnet.GRAD_READY_HOOK = keep_hook
    assert seen and seen[0] != torch.cuda.current_stream(torch.device(device)), 'the bank did not deliver early from a side stream'
    worst = 0.0
    for n, p in model.named_parameters():
        err = (p.grad - want[n]).abs().max().item()
        worst = max(worst, float('inf') if err != err else err / max(1e-6, want[n].abs().max().item()))
    return worst


def check_predictor_graphed_vs_eager(device, steps=3):
    """PredictorTrainer with ``use_graphs`` (forward + backward | clip + update replayed from two hipGraphs, no length read back
    from the device) against the eager trainer on the fixture's batch and weights: every loss of ``steps`` consecutive steps and
    the parameters afterwards (the small predictor has dropout: zeroed here, the masks of the two runs are not aligned)."""
    from msmctts_amd.tasks import build_task
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.trainers.optimizers import build_optimizer
    from msmctts_amd.utils.config import Config
    z = load_npz('small_predictor.npz')
    pc = small_predictor_cfg()
    for key in ('encoder_config', 'decoder_config', 'adaptor_config'):
        pc[key] = dict(pc[key], dropout=0.0)
        if key != 'adaptor_config':
            pc[key]['attn_dropout'] = 0.0
    _, atask = build_small(device)
    batch = {k[len('batch.'):]: t(v).to(device) for k, v in z.items() if k.startswith('batch.')}
    runs = []
    for graphed in (False, True):
        cfg = Config({'id': 'small_predictor_graph', 'task': {'_name': 'MSMCTTS', '_mode': 'train_predictor', 'predictor': pc},
                      'trainer': dict(PREDICTOR_TRAINER, _name='PredictorTrainer'),
                      'optimizer': {'_default': dict(_name='Adam', learning_rate=2e-4, betas=[0.9, 0.98], eps=1e-9, weight_decay=0)},
                      'dataset': dict(samplerate=24000, feature=['mel', 'wav'], frameshift=[300, 1])})
        task = build_task(cfg, mode='train')
        task.load_state_dict({k[len('state.'):]: t(v) for k, v in z.items() if k.startswith('state.')})
        task = task.to(device).train()
        for m in task.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        tr = build_trainer(cfg, task, num_gpus=0, rank=0)
        tr.autoencoder = atask.autoencoder
        tr.optimizer = build_optimizer(task, cfg.optimizer, capturable=True)
        tr.use_graphs = graphed
        logs = []
        for i in range(steps):
            if not tr.replays(i):
                task.zero_grad()
            log = tr.train_step({k: v.clone() for k, v in batch.items()}, i)
            logs.append({k: float(v) for k, v in log['loss'].items()})
        if graphed:
            assert tr._graphs is not None
        runs.append((logs, {k: v.detach().clone() for k, v in task.state_dict().items()}))
    (el, es), (gl, gs) = runs
    for a, b in zip(el, gl):
        assert set(a) == set(b), (sorted(a), sorted(b))
        for k in a:
            assert abs(a[k] - b[k]) <= 2e-3 * max(1.0, abs(a[k])), (k, a[k], b[k])
    for k in es:
        if es[k].dtype.is_floating_point:
            close(gs[k], es[k], 2e-3, 1e-3, k)


def check_triple_loss(device):
    """csrc/losses.hip triple_loss_kernel (Quantize / MultiHeadQuantize.compute_triple_loss on the GPU: reference
    vqgantts/modules.py:86-116, 152-168) against the stock operator chain it replaces: per-frame loss and the gradient of a
    random weighted sum of it, 'sum' and 'mean' reductions, one head and four, predictions near codewords (small hinge sets)
    and far (all codewords active)."""
    from msmctts_amd.networks.vqgantts.modules import MultiHeadQuantize, Quantize
    from msmctts_amd.hip import losses as hiploss
    torch.manual_seed(5)

    def stock(q, p, trg, reduction):
        B, T, D = p.shape
        flat = p.reshape(-1, q.dim)
        dist = (flat.pow(2).sum(1, keepdim=True) - 2 * flat @ q.embed + q.embed.pow(2).sum(0, keepdim=True)).reshape(B, T, -1)
        pos = torch.nn.functional.mse_loss(p, q.embed_code(trg), reduction='none').sum(-1)
        triple = pos.unsqueeze(-1) - dist
        triple = (triple != 0) * (torch.clamp(triple + 1e-6, min=0) / q.dim)
        return triple.mean(-1) if reduction == 'mean' else triple.sum(-1)
    for H, dim, K in ((1, 64, 64), (4, 256, 256), (2, 64, 48)):
        mod = (Quantize(dim, K) if H == 1 else MultiHeadQuantize(dim, K, H)).to(device)
        heads = [mod] if H == 1 else list(mod.quantizers)
        B, T = 3, 37
        trg = torch.randint(0, K, (B, T, H), device=device)
        for spread in (0.05, 3.0):
            near = torch.cat([q.embed_code(trg[..., h]) for h, q in enumerate(heads)], dim=-1)
            p0 = (near + spread * torch.randn(B, T, dim, device=device)).detach()
            wts = torch.rand(B, T, device=device)
            for reduction in ('sum', 'mean'):
                p1 = p0.clone().requires_grad_(True)
                got = mod.compute_triple_loss(p1, trg[..., 0] if H == 1 else trg, reduction=reduction)
                (got * wts).sum().backward()
                p2 = p0.clone().requires_grad_(True)
                want = sum(stock(q, c, trg[..., h], reduction) for h, (q, c) in enumerate(zip(heads, torch.chunk(p2, H, dim=-1)))) / H
                (want * wts).sum().backward()
                scale = max(1e-6, want.abs().max().item())
                # (the target's own codeword enters the hinge as pos - dist = rounding noise of two fp32 routes to one number,
                #  reference and kernel alike: an absolute floor of a few 1e-6 / d)
                assert (got - want).abs().max().item() <= 2e-4 * scale + 5e-6, (H, dim, K, spread, reduction, (got - want).abs().max().item(), scale)
                gscale = max(1e-9, p2.grad.abs().max().item())
                # (a codeword whose hinge argument sits within rounding of zero may be counted by one side only: 1 / (d K) of the scale)
                assert (p1.grad - p2.grad).abs().max().item() <= 2e-3 * gscale + 1e-7, (H, dim, K, spread, reduction)
    assert hiploss.usable(p0)
