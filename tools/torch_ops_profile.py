#!/usr/bin/env python
"""Which stock PyTorch operators (and how many launches) surround the hand-written kernels in one train step."""
import os, sys, random, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd')]
import torch
import bench
from msmctts_amd.synthetic import make_batch
from torch.profiler import profile, ProfilerActivity, record_function


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

# phase labels
for name in ('_segment_a', '_segment_b', '_segment_c'):
    inner = getattr(trainer, name)
    def wrapped(st, inner=inner, name=name):
        with record_function('SEG' + name):
            return inner(st)
    setattr(trainer, name, wrapped)


def step(i):
    trainer.model.zero_grad()
    trainer.optimizer.zero_grad()
    return trainer.train_step(batch, 10 + i)


for i in range(4):
    step(i)
torch.cuda.synchronize()
# host time per segment without GPU sync inside
t0 = time.perf_counter()
for i in range(5):
    step(4 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host-side issue time %.2f ms/step; with final sync %.2f ms/step' % ((t1 - t0) / 5 * 1e3, (t2 - t0) / 5 * 1e3))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(2):
        step(20 + i)
    torch.cuda.synchronize()
out = prof.key_averages().table(sort_by='self_cuda_time_total', row_limit=70, max_name_column_width=60)
open(os.path.join(ROOT, 'gpurun_out', 'torch_ops.txt'), 'w').write(out)
print(out[:200])
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
with open(os.path.join(ROOT, 'gpurun_out', 'torch_ops_count.txt'), 'w') as f:
    for e in rows[:90]:
        f.write('%6d  cpu %8.1f us self  dev %8.1f us  %s\n' % (e.count // 2, e.self_cpu_time_total / 2, getattr(e, 'self_device_time_total', 0) / 2, e.key[:80]))
