#!/usr/bin/env python
"""GPU probe: long hipGraph of stock PyTorch kernels only -- bf16 GEMMs interleaved with multi-block ``sum(dim=0)``
reductions whose scratch (staging buffer + semaphores, zeroed by a memset node) comes from freshly NaN-poisoned pool
blocks.  Counts replays in which any reduction output differs from a reference computed outside the graph.
No kernel of this repository runs here: it separates "stock reduction replayed from a graph" from our code.

    python tools/repro_graph_reduce.py [chain] [replays]
"""
import sys

import torch

dev = torch.device('cuda:0')
torch.manual_seed(0)
CHAIN = int(sys.argv[1]) if len(sys.argv) > 1 else 60
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
shapes = [(1600, 384), (6400, 80), (6400, 384), (1600, 256), (6400, 1024)]
xs = [torch.randn(r, c, device=dev).to(torch.bfloat16) for r, c in shapes]
a = torch.randn(2048, 2048, device=dev).to(torch.bfloat16)
outs = []


def work():
    del outs[:]
    y = a
    for i in range(CHAIN):
        t = [torch.full((n,), float('nan'), device=dev, dtype=torch.bfloat16) for n in (256, 1024, 8192, 1 << 18)]
        del t
        y = (y @ a) * 0.02
        x = xs[i % len(xs)]
        outs.append(x.sum(0))
        outs.append(x.float().sum(0))
    outs.append(y.float().sum())


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    work()
    work()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    work()
g.replay()
torch.cuda.synchronize()
refs = []
for i in range(CHAIN):
    x = xs[i % len(xs)]
    refs.append(x.double().sum(0))
    refs.append(x.double().sum(0))
bad_replays = nonfinite = 0
for rep in range(N):
    for x in xs:                                   # new data every replay: stale results cannot pass
        x.copy_(torch.randn(x.shape, device=dev))
    refs = []
    for i in range(CHAIN):
        x = xs[i % len(xs)]
        refs.append(x.double().sum(0))
        refs.append(x.double().sum(0))
    g.replay()
    torch.cuda.synchronize()
    bad = 0
    for o, r in zip(outs[:-1], refs):
        tol = 0.02 * r.abs() + (1.0 if o.dtype == torch.bfloat16 else 0.05)
        wrong = ~((o.double() - r).abs() <= tol)
        if wrong.any():
            bad += 1
            nonfinite += int((~torch.isfinite(o)).sum())
            if bad_replays < 3 and bad <= 2:
                idx = wrong.nonzero().flatten()[:6].tolist()
                print('replay %d: output %s %s wrong at %s: got %s want %s' % (rep, tuple(o.shape), o.dtype, idx,
                      o.flatten()[idx].tolist(), [round(v, 3) for v in r.flatten()[idx].tolist()]), flush=True)
    bad_replays += bad > 0
print('chain %d, %d replays: %d replays with a wrong reduction, %d non-finite elements' % (CHAIN, N, bad_replays, nonfinite))
