"""One ResBlock1 unit in ONE launch (csrc/resunit.inc, msmc_resunit_forward) against the product's two-launch chain, on the
thin generator stages of the bench configuration (B = 16; C = 32 over 12000 positions, C = 64 over 6000).

    python tools/bench_resunit.py > gpurun_out/resunit_bench.txt

Per (C, k, dilation): µs of the chain (conv_forward with the activation in the producer's epilogue, then conv_forward
with the residual: the tuned variants of the committed table), µs of the fused launch for nt = 2 / 3 / auto, the largest
deviation of y and a between the two.  Last block: the three kernel sizes of one stage as the product issues them (two
grouped calls) against three fused launches back to back."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd')]
import torch  # noqa: E402
from msmctts_amd.hip import conv  # noqa: E402

dev = torch.device('cuda:0')
torch.cuda.set_device(0)
B, SLOPE, ITERS = 16, 0.1, 50


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / ITERS


def case(C, L, k, dil):
    g = torch.Generator().manual_seed(C + k + dil)
    x = torch.randn(B, 1, L, C, generator=g).to(dev, torch.bfloat16)
    w1 = (torch.randn(k, C, C, generator=g) / (C * k) ** 0.5).to(dev, torch.bfloat16)
    w2 = (torch.randn(k, C, C, generator=g) / (C * k) ** 0.5).to(dev, torch.bfloat16)
    b1, b2 = (torch.randn(C, generator=g) * 0.1).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
    g1 = conv.Geometry(1, L, (1, k), (1, 1), (1, dil), (0, dil * (k - 1) // 2), False)
    g2 = conv.Geometry(1, L, (1, k), (1, 1), (1, 1), (0, (k - 1) // 2), False)
    return dict(x=x, w1=w1, w2=w2, b1=b1, b2=b2, g1=g1, g2=g2, k=k, dil=dil)


def chain(c):
    a = conv.conv_forward(c['x'], c['w1'], c['g1'], bias=c['b1'], in_slope=SLOPE, out_slope=SLOPE)
    return a, conv.conv_forward(a, c['w2'], c['g2'], bias=c['b2'], res=c['x'])


def fused(c, nt=0):
    return conv.resunit_forward(c['x'], c['w1'], c['b1'], c['w2'], c['b2'], c['dil'], SLOPE, nt=nt)


print('%-22s %9s %9s %9s %9s   %s' % ('C L k dil', 'chain us', 'nt=2', 'nt=3', 'auto', 'max |dy| / |da| fused vs chain'))
for C, L, ks, dils in ((32, 12000, (3, 7, 11), (1, 3, 5)), (64, 6000, (3,), (1, 3, 5))):
    for k in ks:
        for dil in dils:
            c = case(C, L, k, dil)
            a0, y0 = chain(c)
            a1, y1 = fused(c)
            dy, da = float((y0.float() - y1.float()).abs().max()), float((a0.float() - a1.float()).abs().max())
            row = [timed(lambda: chain(c))] + [timed(lambda nt=nt: fused(c, nt)) for nt in (2, 3, 0)]
            print('%-22s %9.1f %9.1f %9.1f %9.1f   %.3g / %.3g' % ('%d %d %d %d' % (C, L, k, dil), *row, dy, da))
            sys.stdout.flush()

print()
print('one dilation step of a stage as the product issues it: the three kernel sizes in two grouped calls')
for C, L in ((32, 12000), (64, 6000)):
    for dil in (1, 3, 5):
        cs = [case(C, L, k, dil) for k in ((3, 7, 11) if C == 32 else (3,))]

        def grouped():
            a = conv.conv_forward_group([dict(x=c['x'], w=c['w1'], geom=c['g1'], bias=c['b1'], in_slope=SLOPE, out_slope=SLOPE)
                                         for c in cs])
            return conv.conv_forward_group([dict(x=t, w=c['w2'], geom=c['g2'], bias=c['b2'], res=c['x']) for t, c in zip(a, cs)])

        def fused_all():
            return [fused(c) for c in cs]

        print('C %d L %d dil %d (%d branches): grouped chain %.1f us, fused launches %.1f us' %
              (C, L, dil, len(cs), timed(grouped), timed(fused_all)))
        sys.stdout.flush()
