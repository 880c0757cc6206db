#!/usr/bin/env python
"""GPU diagnostic: small golden GAN step, every gradient L2 against the reference fixture, under switch settings."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd'), os.path.join(ROOT, 'tests')]
import torch
import _parity as P
from _util import load_npz, json_field, t
from msmctts_amd.hip import lib, convnet
import msmctts_amd.networks.hifigan.generator as G
import msmctts_amd.networks.hifigan.discriminator as D

dev = torch.device('cuda:0')
z = load_npz('small_steps.npz')
real_make = convnet.make_streams


def run(tag, iteration):
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.trainers.optimizers import build_optimizer
    cfg, task = P.build_small(dev)
    tr = build_trainer(cfg, task, num_gpus=0, rank=0)
    tr.model = task
    tr.optimizer = build_optimizer(task, cfg.optimizer)
    fw = [tuple(int(v) for v in r) for r in z['windows']]
    sw = [(s * 300, e * 300) for s, e in fw]
    tr.random_select = lambda ml: (fw, sw)
    batch = {k[len('batch.'):]: t(v).to(dev) for k, v in z.items() if k.startswith('batch.')}
    snaps = {}
    real_step = tr.optimizer.step

    def spy(names=None):
        key = names[0] if isinstance(names, (list, tuple)) else names
        torch.cuda.synchronize()
        snaps[key] = {n: p.grad.detach().clone() for n, p in task.named_parameters()
                      if n.startswith(key + '.') and p.grad is not None}
        return real_step(names)
    real_clip_step = tr.optimizer.clip_and_step

    def clip_spy(name, max_norm):               # the fused clip + AdamW: gradients are clipped in place, as clip_grad_norm_ does
        out = real_clip_step(name, max_norm)
        torch.cuda.synchronize()
        snaps[name] = {n: p.grad.detach().float().cpu().clone() for n, p in task.named_parameters()
                       if n.startswith(name + '.') and p.grad is not None}
        return out
    tr.optimizer.step = spy
    tr.optimizer.clip_and_step = clip_spy
    task.zero_grad()
    tr.train_step(batch, iteration)
    bad = []
    for child, gd in snaps.items():
        names = json_field(z['%s.grad_names.%s' % (tag, child)])
        for n, w in zip(names, z['%s.grad_l2.%s' % (tag, child)]):
            g = gd[n].double().norm().item()
            if abs(g - w) > 2e-3 * max(w, 1e-3) + 1e-6:
                bad.append((n, g / max(w, 1e-12)))
    return bad


L = lib.get()
from collections import Counter
for label, gs, ds, pipe, narrow in [('both wide', 1, 1, 1, 0)] * 14 + [('both', 1, 1, 1, 1)] * 10:
    G.make_streams = real_make if gs else (lambda device, n: [])
    D.make_streams = real_make if ds else (lambda device, n: [])
    L.msmc_conv_set_pipeline(pipe)
    L.msmc_conv_set_narrow(narrow)
    for tag, it in (('warm', 0), ('gan', 6)):
        bad = run(tag, it)
        cnt = Counter('.'.join(n.split('.')[:2]) for n, r in bad)
        print('%-10s %-4s mismatches %3d %s | %s' % (label, tag, len(bad), dict(cnt), ' '.join('%s=%.3f' % (n.split('autoencoder.')[-1], r) for n, r in bad[:6])), flush=True)
