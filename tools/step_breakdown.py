#!/usr/bin/env python
"""Per-layer time table of one GAN-phase train step (bench configuration), kernels timed in isolation:
side streams are disabled so every launch runs alone between two HIP events.  Output: gpurun_out/step_breakdown.txt"""
import os, sys, random, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd')]
import torch
import bench
from msmctts_amd.hip import conv as hipconv, convnet, vq as hipvq, spectral, losses
import msmctts_amd.networks.hifigan.generator as G
import msmctts_amd.networks.hifigan.discriminator as D
from msmctts_amd.synthetic import make_batch

if os.environ.get('SERIAL', '1') == '1':
    G.make_streams = D.make_streams = lambda device, n: []


class A(object):
    codewords, heads, batch, frames, graph, dtype, no_autocast = 256, 4, 16, 400, False, 'bf16', False


from msmctts_amd.hip import lib as _lib
if os.environ.get('GATHER_GEN'):
    _lib.get().msmc_conv_set_gather_generation(int(os.environ['GATHER_GEN']))
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
cfg, trainer = bench.build(A, dev, 0, 1)
batch = make_batch(A.batch, A.frames, 80, 300, seed=1234, rank=0, device='cpu')
lengths = batch['mel_length'].tolist()
batch = {k: v.to(dev) for k, v in batch.items()}
batch['mel_length_host'] = lengths
trainer.rng = random.Random(1234)
recs = collections.defaultdict(list)
enabled = [False]


def wrap(mod, fn, label, keyf):
    inner = getattr(mod, fn)

    def timed(*a, **k):
        if not enabled[0]:
            return inner(*a, **k)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = inner(*a, **k)
        e.record()
        recs[(label,) + keyf(*a, **k)].append((s, e))
        return out
    setattr(mod, fn, timed)


def sh(t):
    return 'x'.join(str(int(v)) for v in t.shape)


wrap(hipconv, 'conv_forward', 'fwd', lambda x, w, geom, *a, **k: (sh(x), sh(w), str(geom.stride) + str(geom.dil) if hasattr(geom, 'dil') else ''))
wrap(hipconv, 'conv_dgrad', 'dgrad', lambda g, w, geom, *a, **k: (sh(g), sh(w), ''))
wrap(hipconv, 'conv_wgrad', 'wgrad', lambda x, g, geom, n, *a, **k: (sh(x), sh(g), str(n)))
wrap(hipconv, 'conv_transpose1d_forward', 'convT', lambda x, w, *a, **k: (sh(x), sh(w), ''))
wrap(hipconv, 'conv_transpose1d_dgrad', 'convT dgrad', lambda g, w, *a, **k: (sh(g), sh(w), ''))
wrap(hipconv, 'conv_transpose1d_wgrad', 'convT wgrad', lambda x, g, *a, **k: (sh(x), sh(g), ''))
wrap(hipconv, 'conv_forward_group', 'fwd-group', lambda items: (sh(items[0]['x']), sh(items[0]['w']), 'n=%d' % len(items)))
wrap(hipconv, 'conv_dgrad_group', 'dgrad-group', lambda items: (sh(items[0]['g']), sh(items[0]['wb']), 'n=%d' % len(items)))
wrap(hipconv, 'conv_wgrad_group', 'wgrad-group', lambda items: (sh(items[0]['x']), sh(items[0]['g']), 'n=%d' % len(items)))
wrap(hipconv, 'reflect_fold', 'reflect_fold', lambda g, *a, **k: (sh(g), '', ''))
wrap(hipconv, 'lrelu_bwd', 'lrelu_bwd', lambda g, *a, **k: (sh(g), '', ''))
wrap(hipconv, 'colsum', 'colsum', lambda g, *a, **k: (sh(g), '', ''))


def step(i):
    trainer.model.zero_grad()
    trainer.optimizer.zero_grad()
    return trainer.train_step(batch, 10 + i)


for i in range(3):
    step(i)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(5):
    step(3 + i)
torch.cuda.synchronize()
plain = (time.perf_counter() - t0) / 5 * 1e3
enabled[0] = True
N = 3
for i in range(N):
    step(10 + i)
torch.cuda.synchronize()
enabled[0] = False
rows = []
for key, ev in recs.items():
    us = [s.elapsed_time(e) * 1e3 for s, e in ev]
    rows.append((sum(us) / N, len(us) / N, sum(us) / len(us), key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
with open(os.path.join(ROOT, 'gpurun_out', os.environ.get('BREAKDOWN_OUT', 'step_breakdown.txt')), 'w') as f:
    f.write('step (no events) %.2f ms; hand-written conv-family launches: %.2f ms/step over %d launches\n'
            % (plain, tot / 1e3, sum(r[1] for r in rows)))
    by = collections.defaultdict(float)
    for r in rows:
        by[r[3][0]] += r[0]
    f.write('  '.join('%s %.2f ms' % kv for kv in sorted(by.items(), key=lambda kv: -kv[1])) + '\n')
    for tot_us, n, avg, key in rows:
        f.write('%8.1f us/step  n=%5.1f  avg %7.1f us  %s\n' % (tot_us, n, avg, ' '.join(key)))
print(open(os.path.join(ROOT, 'gpurun_out', os.environ.get('BREAKDOWN_OUT', 'step_breakdown.txt'))).read()[:400])
from msmctts_amd.hip import conv as _c
import json as _json
with open(os.path.join(ROOT, 'gpurun_out', 'tuned.json'), 'w') as f:
    _json.dump([dict(kind=k[0], sig=list(map(str, k[1:])), variant=v[0], shift=v[1], times={str(c): t for c, t in v[2].items()}) for k, v in _c.TUNED.items()], f, indent=0)
wins = collections.Counter((k[0], v[0], v[1]) for k, v in _c.TUNED.items())
print('tuned choices (kind, variant, split_shift): count', dict(wins))
