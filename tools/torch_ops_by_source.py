#!/usr/bin/env python
"""Which source lines of this package issue the stock PyTorch kernels of one train step (count and device time)."""
import os, sys, random, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd')]
import torch
import bench
from msmctts_amd.synthetic import make_batch
from torch.profiler import profile, ProfilerActivity


class A(object):
    codewords, heads, batch, frames, graph, dtype, no_autocast = 256, 4, 16, 400, False, 'bf16', False


os.environ['MSMC_STREAMS'] = '0'; os.environ['MSMC_AUTOTUNE'] = '0'
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
cfg, trainer = bench.build(A, dev, 0, 1)
batch = make_batch(A.batch, A.frames, 80, 300, seed=1234, rank=0, device='cpu')
lengths = batch['mel_length'].tolist()
batch = {k: v.to(dev) for k, v in batch.items()}
batch['mel_length_host'] = lengths
trainer.rng = random.Random(1234)


def step(i):
    trainer.model.zero_grad()
    trainer.optimizer.zero_grad()
    return trainer.train_step(batch, 10 + i)


for i in range(4):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step(20)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for ev in prof.events():
    ks = getattr(ev, 'kernels', None) or []
    dt = sum(k.duration for k in ks)
    if dt <= 0:
        continue
    src = None
    for fr in (ev.stack or []):
        if '/msmc-tts_amd/' in fr or '/oracle/' in fr:
            src = fr.split('msmc-tts_amd/')[-1]
            break
    if src is None:
        src = 'torch: ' + (ev.stack[0][-60:] if ev.stack else ev.name)
    a = agg[src]
    a[0] += 1
    a[1] += dt
    a[2][ev.name[:40]] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
with open(os.path.join(ROOT, 'gpurun_out', 'torch_ops_by_source.txt'), 'w') as f:
    tot = sum(v[1] for v in agg.values())
    f.write('device time attributed: %.2f ms over %d op events\n' % (tot / 1e3, sum(v[0] for v in agg.values())))
    for src, (n, dt, names) in rows[:70]:
        f.write('%7.1f us  n=%4d  %-70s %s\n' % (dt, n, src[:70], dict(names.most_common(3))))
print(open(os.path.join(ROOT, 'gpurun_out', 'torch_ops_by_source.txt')).read()[:9000])

want = ('aten::copy_', 'aten::add_', 'aten::sum', 'aten::fill_', 'aten::mul', 'aten::add', 'aten::div')
agg2 = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    ks = getattr(ev, 'kernels', None) or []
    dt = sum(k.duration for k in ks)
    if dt <= 0 or ev.name not in want:
        continue
    frames = [fr for fr in (ev.stack or []) if '/msmc-tts_amd/' in fr or 'bench.py' in fr or '/torch/optim' in fr or 'clip_grad' in fr or 'autograd' in fr]
    key = (ev.name, frames[0].split('/')[-1][:80] if frames else (ev.stack[0][-60:] if ev.stack else '?'))
    agg2[key][0] += 1
    agg2[key][1] += dt
with open(os.path.join(ROOT, 'gpurun_out', 'torch_small_ops_by_line.txt'), 'w') as f:
    for key, (n, dt) in sorted(agg2.items(), key=lambda kv: -kv[1][1])[:80]:
        f.write('%7.1f us  n=%4d  %-12s %s\n' % (dt, n, key[0], key[1]))
print(open(os.path.join(ROOT, 'gpurun_out', 'torch_small_ops_by_line.txt')).read()[:7000])
