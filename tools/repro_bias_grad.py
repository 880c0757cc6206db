#!/usr/bin/env python
"""GPU probe: does a stock bf16 ``sum(dim=0)`` (the bias gradient of an autocast nn.Linear) ever return a wrong /
non-finite element when replayed from a hipGraph right after kernels that leave NaN patterns in the graph pool?

Background (round 2 NaN hunt): the benchmarked bf16 step, replayed as hipGraphs, occasionally produced ONE NaN element
in the bias gradient of a stock ``nn.Linear`` (mel_predictor.bias, slf_attn.linear.bias); clipping then spread it.
"""
import sys

import torch
import torch.nn.functional as F

dev = torch.device('cuda:0')
torch.manual_seed(0)
R, C = int(sys.argv[1]) if len(sys.argv) > 1 else 6400, int(sys.argv[2]) if len(sys.argv) > 2 else 384
N = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
x = torch.randn(R, 256, device=dev)
w = torch.randn(C, 256, device=dev, requires_grad=True)
b = torch.randn(C, device=dev, requires_grad=True)
go = torch.randn(R, C, device=dev)


def work():
    # poison: temporaries full of NaN, freed right before the linear's backward allocates its buffers
    t = [torch.full((n,), float('nan'), device=dev, dtype=torch.bfloat16) for n in (512, 2048, 65536, 1 << 20)]
    del t
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = F.linear(x, w, b)
    (y.float() * go).sum().backward()


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        w.grad = b.grad = None
        work()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
w.grad = b.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    work()
ref = go.to(torch.bfloat16).double().sum(0)
bad = worst = 0
for i in range(N):
    go.copy_(torch.randn(R, C, device=dev))
    ref = go.to(torch.bfloat16).double().sum(0)
    g.replay()
    err = (b.grad.double() - ref).abs() / (ref.abs() + 10.0)
    nf = int((~torch.isfinite(b.grad)).sum())
    e = float(err[torch.isfinite(err)].max())
    worst = max(worst, e)
    if nf or e > 0.05:
        bad += 1
        if bad <= 5:
            print('replay %d: %d non-finite, max rel err %.3g' % (i, nf, e), flush=True)
print('graph replays %d: bad %d, worst finite rel err %.3g' % (N, bad, worst))
# the same eagerly
bad = 0
for i in range(N // 4):
    w.grad = b.grad = None
    work()
    if int((~torch.isfinite(b.grad)).sum()):
        bad += 1
print('eager runs %d: bad %d' % (N // 4, bad))
