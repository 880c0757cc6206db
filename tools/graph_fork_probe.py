"""Does a hipGraph captured with a fork/join onto a second stream run the two branches concurrently on gfx950?

Two independent chains of N dependent launches each, captured (a) back to back on one stream and (b) forked onto
two streams; replay time of each, for launches that fill a fraction of the chip and for launches that fill it.
Usage: python tools/graph_fork_probe.py   (prints one table; run it under the same environment as bench.py)
"""
import os, sys, time
import torch


def chain(x, w, n):
    for _ in range(n):
        x = torch.mm(x, w)
    return x


def ew_chain(x, n):
    for _ in range(n):
        x = x * 1.0001 + 0.5
    return x


def timed(g, reps=20):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def capture(fn_a, fn_b, fork):
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    main = torch.cuda.Stream()
    with torch.cuda.stream(main):
        fn_a(); fn_b()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=main):
            if fork:
                ev = torch.cuda.Event(); ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    fn_b()
                    ev2 = torch.cuda.Event(); ev2.record(side)
                fn_a()
                main.wait_event(ev2)
            else:
                fn_a(); fn_b()
    return g


def main():
    dev = 'cuda'
    n = 60
    print('env DEBUG_CLR_GRAPH_PACKET_CAPTURE =', os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'))
    for name, mk in (
        ('mm 256x256x256 bf16 (few blocks)', lambda: (torch.randn(256, 256, device=dev, dtype=torch.bfloat16), torch.eye(256, device=dev, dtype=torch.bfloat16))),
        ('mm 2048x512x512 bf16', lambda: (torch.randn(2048, 512, device=dev, dtype=torch.bfloat16), torch.eye(512, device=dev, dtype=torch.bfloat16))),
        ('mm 16384x512x512 bf16 (fills chip)', lambda: (torch.randn(16384, 512, device=dev, dtype=torch.bfloat16), torch.eye(512, device=dev, dtype=torch.bfloat16))),
    ):
        xa, wa = mk(); xb, wb = mk()
        fa = lambda: chain(xa, wa, n)
        fb = lambda: chain(xb, wb, n)
        t1 = timed(capture(fa, fb, False)); t2 = timed(capture(fa, fb, True))
        print('%-40s serial %.3f ms  forked %.3f ms  (%.2fx)' % (name, t1, t2, t1 / t2))
    for name, numel in (('elementwise 64K', 1 << 16), ('elementwise 4M', 1 << 22), ('elementwise 64M', 1 << 26)):
        xa = torch.randn(numel, device=dev); xb = torch.randn(numel, device=dev)
        fa = lambda: ew_chain(xa, n)
        fb = lambda: ew_chain(xb, n)
        t1 = timed(capture(fa, fb, False)); t2 = timed(capture(fa, fb, True))
        print('%-40s serial %.3f ms  forked %.3f ms  (%.2fx)' % (name, t1, t2, t1 / t2))


if __name__ == '__main__':
    main()
