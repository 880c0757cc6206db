#!/usr/bin/env python
"""List host<->device synchronisation points inside one eager train step (torch sync debug mode)."""
import os, sys, random, warnings, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd')]
import torch
import bench
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
seen = {}
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if 'find_syncs' not in f.filename and 'warnings' not in f.filename]
    key = tuple((os.path.basename(f.filename), f.lineno, f.name) for f in st[-6:])
    seen[key] = seen.get(key, 0) + 1
warnings.showwarning = showwarning
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
trainer.model.zero_grad(); trainer.optimizer.zero_grad()
trainer.train_step(batch, 20)
torch.cuda.set_sync_debug_mode('default')
for k, v in seen.items():
    print(v, k)
print('total sync points per step:', sum(seen.values()))
