#!/usr/bin/env python
"""Hand-written conv-family kernel launches of one eager train step (msmc_conv_launch_count)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd')]
import torch
import bench
from msmctts_amd.hip import lib
from msmctts_amd.synthetic import make_batch


class A(object):
    codewords, heads, batch, frames, graph, dtype, no_autocast = 256, 4, 16, 400, False, 'bf16', False


dev = torch.device('cuda:0')
torch.cuda.set_device(0)
cfg, trainer = bench.build(A, dev, 0, 1)
batch = make_batch(A.batch, A.frames, 80, 300, seed=1234, rank=0, device='cpu')
lengths = batch['mel_length'].tolist()
batch = {k: v.to(dev) for k, v in batch.items()}
batch['mel_length_host'] = lengths
trainer.rng = random.Random(1234)
for i in range(3):
    trainer.model.zero_grad(); trainer.optimizer.zero_grad()
    trainer.train_step(batch, 10 + i)
torch.cuda.synchronize()
# the autograd engine launches from its own thread: count per thread is thread-local, so count through a wrapper
import collections, ctypes
L = lib.get()
names = collections.Counter()
calls = [0]
for fn in ('msmc_conv_gather', 'msmc_conv_wgrad', 'msmc_conv_gather_group', 'msmc_conv_wgrad_group'):
    inner = getattr(L, fn)
    def wrapped(*a, _inner=inner, _fn=fn):
        n0 = L.msmc_conv_launch_count()
        r = _inner(*a)
        k = L.msmc_conv_launch_count() - n0
        calls[0] += k
        names[(_fn, L.msmc_conv_last_kernel().decode().split('<')[0], k)] += 1
        return r
    setattr(L, fn, wrapped)
trainer.model.zero_grad(); trainer.optimizer.zero_grad()
trainer.train_step(batch, 20)
torch.cuda.synchronize()
print('conv-family kernel launches per step:', calls[0])
for k, v in sorted(names.items(), key=lambda kv: -kv[1]):
    print(v, k)
