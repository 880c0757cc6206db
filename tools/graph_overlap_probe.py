#!/usr/bin/env python
"""Do two INDEPENDENT chains of medium kernels (grids that under-fill the 256 CUs) overlap when issued on two streams --
eagerly and as two branches of one hipGraph?  (tools/graph_concurrency_probe.py asks the same of tiny kernels.)"""
import sys, time
sys.path[:0] = ['.', 'msmc-tts_amd']
import msmctts_amd  # noqa  (runtime switches before the first HIP call)
import torch
dev = torch.device('cuda:0')
NCH = int(sys.argv[1]) if len(sys.argv) > 1 else 2
LEN = 40
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
a = [torch.randn(M, 512, device=dev, dtype=torch.bfloat16) for _ in range(NCH)]
w = [torch.randn(512, 512, device=dev, dtype=torch.bfloat16) / 22 for _ in range(NCH)]
streams = [torch.cuda.Stream() for _ in range(NCH)]


def chain(x, wt):
    for _ in range(LEN):
        x = x @ wt
    return x


def serial():
    return [chain(x, wt) for x, wt in zip(a, w)]


def forked():
    main = torch.cuda.current_stream()
    outs = []
    for x, wt, st in zip(a, w, streams):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            outs.append(chain(x, wt))
    for st in streams:
        main.wait_stream(st)
    return outs


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print('chains %d  rows %d  (%d GEMMs of %dx512x512 each)' % (NCH, M, LEN, M))
print('serial eager   %.3f ms' % timeit(serial))
print('forked eager   %.3f ms' % timeit(forked))
side = torch.cuda.Stream()
for name, fn in (('serial', serial), ('forked', forked)):
    g = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        keep = fn()
    print('%s graph   %.3f ms' % (name, timeit(g.replay)))
