#!/usr/bin/env python
"""Do hipGraph branches run concurrently?  10 independent chains of 20 small dependent kernels: serial on one
stream, forked over 10 streams (eager), and the forked version captured into a graph and replayed."""
import time, torch
dev = torch.device('cuda:0')
NCH, LEN = 10, 20
xs = [torch.randn(64 * 1024, device=dev) for _ in range(NCH)]          # small kernels: latency-bound, few workgroups
streams = [torch.cuda.Stream() for _ in range(NCH)]


def chain(x):
    for _ in range(LEN):
        x = torch.sin(x) * 1.0001
    return x


def serial():
    return [chain(x) for x in xs]


def forked():
    main = torch.cuda.current_stream()
    outs = []
    for x, st in zip(xs, streams):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            outs.append(chain(x))
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
# bigger kernels (fill the GPU): 16M elements
xs = [torch.randn(16 * 1024 * 1024, device=dev) for _ in range(NCH)]
print('big serial eager %.3f ms, forked eager %.3f ms' % (timeit(serial, 5), timeit(forked, 5)))
