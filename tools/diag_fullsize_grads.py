#!/usr/bin/env python
"""GPU diagnostic: fp32 product GAN step vs the oracle at CSMSC size -- per-parameter gradient error table
(relative L2 error of the gradient TENSOR, not only of its norm), D step and G step."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd'), os.path.join(ROOT, 'tests')]
import msmctts_amd  # noqa
import torch
import test_gpu_fullsize as T
from oracle import model as omodel
from oracle.step import OracleTrainer

omodel.RESSTACK_DROPOUT = 0.0
name = sys.argv[1] if len(sys.argv) > 1 else 'config2'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kw = T.CONFIGS[name]
cfg = T._cfg(B, dropout=False, **kw)
tr = T._build(cfg, dropout=False)
task = tr.model
cpu_batch, batch = T._batch(B, 400, kw.get('in_dim', 80))
r = random.Random(7)
fw = []
for n in batch['mel_length_host']:
    s = r.randrange(max(1, n - 40))
    fw.append((s, s + 40))
sw = [(s * 300, e * 300) for s, e in fw]
state0 = {k: v.detach().float().cpu().clone() for k, v in task.state_dict().items()}
tcfg = {k: v for k, v in cfg.trainer.to_dict().items() if k != '_name'}
oracle = OracleTrainer(state0, cfg.task.to_dict(), tcfg)
tr.random_select = lambda ml: (fw, sw)
snaps = {}
real_step = tr.optimizer.step
real_clip = torch.nn.utils.clip_grad_norm_


def spy(names=None):
    key = names[0] if isinstance(names, (list, tuple)) else names
    torch.cuda.synchronize()
    if key == 'discriminator':
        snaps[key] = {n: p.grad.detach().float().cpu().clone() for n, p in task.named_parameters()
                      if n.startswith(key + '.') and p.grad is not None}
    return real_step(names)


def clip_spy(params, thresh):
    params = list(params)
    torch.cuda.synchronize()
    snaps['autoencoder'] = {n: p.grad.detach().float().cpu().clone() for n, p in task.named_parameters()
                            if n.startswith('autoencoder.') and p.grad is not None}
    tot = sum(float(v.double().pow(2).sum()) for v in snaps['autoencoder'].values()) ** 0.5
    out = real_clip(params, thresh)
    tot2 = sum(float(p.grad.double().pow(2).sum()) for p in params if p.grad is not None) ** 0.5
    print('clip_spy: snapshot norm %.6g, real_clip returned %.6g, norm after %.6g, nparams %d / %d snap' % (tot, float(out), tot2, len(params), len(snaps['autoencoder'])))
    return out


tr.optimizer.step = spy
import msmctts_amd.trainers.msmctts_trainer as MT
MT.nn.utils.clip_grad_norm_ = clip_spy
task.zero_grad()
log = tr.train_step(batch, 10)
MT.nn.utils.clip_grad_norm_ = real_clip
keep = {}
ref = oracle.train_step(dict(cpu_batch), 10, windows=(fw, sw), keep=keep)
print('losses', {k: (round(float(v), 5), round(ref['loss'][k], 5)) for k, v in log['loss'].items()})
print('grad_norm', float(tr.grad_norm), keep['grad_norm'])
for child, okey in (('discriminator', 'd_grads'), ('autoencoder', 'g_grads')):
    rows = []
    for n, g in snaps[child].items():
        if n not in keep[okey]:
            continue
        w = keep[okey][n].double()
        rows.append((float((g.double() - w).norm() / (w.norm() + 1e-12)), float(g.double().norm()), float(w.norm()), n))
    rows.sort(reverse=True)
    print('== %s: %d tensors, worst relative L2 errors' % (child, len(rows)))
    for e, m, w, n in rows[:25]:
        print('  %.3e  got %.5g  oracle %.5g  %s' % (e, m, w, n))
    print('  median %.3e' % sorted(r_[0] for r_ in rows)[len(rows) // 2])
