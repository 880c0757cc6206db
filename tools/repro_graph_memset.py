#!/usr/bin/env python
"""GPU probe: is a hipMemsetAsync node inside a captured hipGraph ordered against its neighbours on replay?

Graph (one stream):  for i in chain:  fill(S_i, garbage) -> hipMemsetAsync(S_i, 0, nbytes) -> R_i = S_i.clone()
Every R_i must be all zero on every replay.  PyTorch's multi-block reductions zero their semaphores exactly so
(cudaMemsetAsync between the allocation and the reduce kernel), which is how this was found: with ROCm 7.2's default
graph path (AQL packet capture) the bf16 train step replayed from hipGraphs got sporadic stale / NaN bias gradients.

    python tools/repro_graph_memset.py [chain] [replays] [nbytes]
"""
import ctypes
import os
import sys

import torch

if os.environ.get('LATE_ENV'):          # is the knob still honoured when set after ``import torch`` (before the first HIP call)?
    os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = '0'

dev = torch.device('cuda:0')
CHAIN = int(sys.argv[1]) if len(sys.argv) > 1 else 50
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
NBYTES = int(sys.argv[3]) if len(sys.argv) > 3 else 64
hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
S = [torch.empty(NBYTES // 4, device=dev) for _ in range(CHAIN)]
a = torch.randn(1024, 1024, device=dev)
R = []


def work():
    del R[:]
    y = a
    for s in S:
        s.fill_(float('nan'))
        y = (y @ a) * 0.03
        rc = hip.hipMemsetAsync(s.data_ptr(), 0, NBYTES, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        R.append(s.clone())


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    work()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
eager_bad = sum(int((r != 0).sum()) for r in R)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    work()
bad_replays = bad_elems = 0
for rep in range(N):
    g.replay()
    torch.cuda.synchronize()
    nb = sum(int((r != 0).sum() + torch.isnan(r).sum()) for r in R)
    bad_replays += nb > 0
    bad_elems += nb
# memcpy nodes: a counter kernel, a broadcast kernel, then clone() = hipMemcpyAsync D2D; errs counts stale copies
c = torch.zeros(1, device=dev)
errs = torch.zeros(1, device=dev)
T = [torch.empty(1024, device=dev) for _ in range(CHAIN)]


def work2():
    y = a
    for t in T:
        c.add_(1.0)
        t.copy_(c.expand_as(t))
        y = (y @ a) * 0.03
        r = t.clone()
        errs.add_((r != c).sum())


with torch.cuda.stream(side):
    work2()
torch.cuda.synchronize()
e0 = float(errs)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=side):
    work2()
for rep in range(N):
    g2.replay()
torch.cuda.synchronize()
print('memcpy nodes: eager stale %d; graph stale elements over %d replays: %d' % (e0, N, float(errs) - e0))
print('memset %d B, chain %d: eager bad elements %d; graph: %d of %d replays saw non-zero data after the memset '
      '(%d elements)' % (NBYTES, CHAIN, eager_bad, bad_replays, N, bad_elems))
