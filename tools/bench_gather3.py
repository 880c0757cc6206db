#!/usr/bin/env python
"""GPU microbench: forward convolutions of the bench configuration -- the tuned second-generation choice against every
third-generation configuration (desc.variant 16..23; 24..31 = the same with the halo tile by LDS-DMA) and the persistent
thin-layer kernel (32; 33 = its deferred-epilogue form), checked against the second-generation output.  VARIANTS="24 25 32" selects the columns.

    python tools/bench_gather3.py [filter]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd')]
import msmctts_amd  # noqa
import torch
from msmctts_amd.hip import conv, lib

dev = torch.device('cuda:0')
# (name, B, Cin, Cout, H, W, kernel, stride, dilation, padding, reflect, in_slope)
SHAPES = [
    ('ffn w1 T400 256->1024 k3', 16, 256, 1024, 1, 400, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0),
    ('ffn w2 T400 1024->256 k3', 16, 1024, 256, 1, 400, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0),
    ('ffn w1 T100 256->1024 k3', 16, 256, 1024, 1, 100, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0),
    ('ffn w2 T100 1024->256 k3', 16, 1024, 256, 1, 100, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0),
    # the predictor's FFT blocks at B = 64 (configs.am_config): channel counts that end inside a tile of 64
    ('am w1 T400 600->1536 k3', 64, 600, 1536, 1, 400, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0),
    ('am w2 T400 1536->600 k3', 64, 1536, 600, 1, 400, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0),
    ('rb C256 L240 k3', 16, 256, 256, 1, 240, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.1),
    ('rb C256 L240 k11 d5', 16, 256, 256, 1, 240, (1, 11), (1, 1), (1, 5), (0, 25), False, 0.1),
    ('rb C128 L1200 k3', 16, 128, 128, 1, 1200, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.1),
    ('rb C128 L1200 k7 d3', 16, 128, 128, 1, 1200, (1, 7), (1, 1), (1, 3), (0, 9), False, 0.1),
    ('rb C128 L1200 k11', 16, 128, 128, 1, 1200, (1, 11), (1, 1), (1, 1), (0, 5), False, 0.1),
    ('rb C64 L6000 k3', 16, 64, 64, 1, 6000, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.1),
    ('rb C64 L6000 k7', 16, 64, 64, 1, 6000, (1, 7), (1, 1), (1, 1), (0, 3), False, 0.1),
    ('rb C64 L6000 k11 d5', 16, 64, 64, 1, 6000, (1, 11), (1, 1), (1, 5), (0, 25), False, 0.1),
    ('rb C32 L12000 k3', 16, 32, 32, 1, 12000, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.1),
    ('rb C32 L12000 k11', 16, 32, 32, 1, 12000, (1, 11), (1, 1), (1, 1), (0, 5), False, 0.1),
    ('conv_pre 256->512 k7', 16, 256, 512, 1, 40, (1, 7), (1, 1), (1, 1), (0, 3), False, 1.0),
    ('mpd p2 256->512 s3', 32, 256, 512, 223, 2, (5, 1), (3, 1), (1, 1), (2, 0), False, 0.2),
    ('mpd p2 512->512 s1', 32, 512, 512, 75, 2, (5, 1), (1, 1), (1, 1), (2, 0), False, 0.2),
    ('mpd p11 512->512 s1', 32, 512, 512, 14, 11, (5, 1), (1, 1), (1, 1), (2, 0), False, 0.2),
    ('mpd p2 64->256 s3', 32, 64, 256, 667, 2, (5, 1), (3, 1), (1, 1), (2, 0), False, 0.2),
    ('mrd h240 64->128 s2', 32, 64, 128, 241, 26, (3, 3), (2, 2), (1, 1), (1, 1), True, 0.2),
    ('mrd h240 128->256 s1', 32, 128, 256, 121, 13, (3, 3), (1, 1), (1, 1), (1, 1), True, 0.2),
    ('mrd h240 256->512 s2', 32, 256, 512, 121, 13, (3, 3), (2, 2), (1, 1), (1, 1), True, 0.2),
    ('mrd h50 64->128 s1', 32, 64, 128, 51, 121, (3, 3), (1, 1), (1, 1), (1, 1), True, 0.2),
    # thin discriminator layers (variant 50 / the direct kernels): run with  VARIANTS="50 2 3 8" ... thin
    ('thin mrd 4->8 s2 H31 W801', 32, 4, 8, 31, 801, (3, 3), (2, 2), (1, 1), (1, 1), True, 1.0),
    ('thin mrd 8->16 s1 H16 W401', 32, 8, 16, 16, 401, (3, 3), (1, 1), (1, 1), (1, 1), True, 1.0),
    ('thin mrd 16->32 s2 H16 W401', 32, 16, 32, 16, 401, (3, 3), (2, 2), (1, 1), (1, 1), True, 1.0),
    ('thin mrd 32->64 s1 H8 W201', 32, 32, 64, 8, 201, (3, 3), (1, 1), (1, 1), (1, 1), True, 1.0),
    ('thin mrd 16->32 s2 H121 W51', 32, 16, 32, 121, 51, (3, 3), (2, 2), (1, 1), (1, 1), True, 1.0),
    ('thin mpd p2 16->64 s3', 32, 16, 64, 2000, 2, (5, 1), (3, 1), (1, 1), (2, 0), False, 0.2),
    ('thin mpd p11 16->64 s3', 32, 16, 64, 364, 11, (5, 1), (3, 1), (1, 1), (2, 0), False, 0.2),
]
flt = sys.argv[1] if len(sys.argv) > 1 else ''
if os.environ.get('MSMC_PROBE_LIB'):       # a sensitivity build of the library (tools/r04_pad_probe.sh)
    lib._lib = lib.load(os.environ['MSMC_PROBE_LIB'])
L = lib.get()


def timed(desc, stream, iters=20):
    rc = L.msmc_conv_gather(ctypes.byref(desc), stream)
    if rc != 0:
        return None
    for _ in range(2):
        L.msmc_conv_gather(ctypes.byref(desc), stream)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(3e7))             # park the GPU: the launches below queue up, the events bracket execution only
    s.record()
    for _ in range(iters):
        L.msmc_conv_gather(ctypes.byref(desc), stream)
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters * 1e3


VARIANTS = [int(v) for v in os.environ.get('VARIANTS', ' '.join(str(v) for v in range(16, 34))).split()]
print('%-28s %9s %7s | %s' % ('layer', 'GFLOP', 'gen2 us', ' '.join('v%-6d' % v for v in VARIANTS)))
for name, B, Cin, Cout, H, W, k, s_, dil, pad, reflect, slope in SHAPES:
    if flt not in name:
        continue
    torch.manual_seed(0)
    geom = conv.Geometry(H, W, k, s_, dil, pad, reflect)
    T = k[0] * k[1]
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
    w = (torch.randn(T, Cout, Cin, device=dev) / (Cin * T) ** 0.5).bfloat16()
    bias = torch.randn(Cout, device=dev)
    res = torch.randn(B, geom.Hout, geom.Wout, Cout, device=dev).bfloat16()
    ref = conv.conv_forward(x, w, geom, bias=bias, in_slope=slope, res=res)       # tuned / autotuned generation <= 2
    desc, out = conv._forward_desc(x, w, geom, bias, slope, res, None, 1.0, 1.0)
    stream = lib.stream(x)
    base_variant = desc.variant
    t2 = timed(desc, stream)
    gflop = 2.0 * B * geom.Hout * geom.Wout * Cout * Cin * T / 1e9
    cells = []
    best = (t2, base_variant)
    for v in VARIANTS:
        d3 = lib.ConvDesc.from_buffer_copy(desc)
        d3.variant = v
        out.zero_()
        t3 = timed(d3, stream)
        if t3 is None:
            cells.append('   -   ')
            continue
        err = (out.float() - ref.float()).abs().max().item() / max(1e-6, ref.float().abs().max().item())
        cells.append('%6.1f%s' % (t3, ' ' if err < 1e-2 else '!'))
        if err >= 1e-2:
            cells[-1] = 'ERR%.0e' % err
        elif t3 < best[0]:
            best = (t3, v)
    if os.environ.get('ABLATE'):
        for v in [int(a) for a in os.environ['ABLATE'].split()]:
            row = []
            for abl in [int(a) for a in os.environ.get('ABLS', '0 1 2 4 8 9 6 7 15').split()]:
                d3 = lib.ConvDesc.from_buffer_copy(desc)
                d3.variant, d3.split_shift = v, abl
                t3 = timed(d3, stream)
                row.append('abl%-2d %s' % (abl, '%.1f' % t3 if t3 else '-'))
            print('%-28s v%d: %s' % (name, v, '  '.join(row)), flush=True)
        continue
    print('%-28s %9.2f %7.1f | %s | best v%d %.1f us = %.1f TF/s (x%.2f)' % (name, gflop, t2, ' '.join(cells), best[1], best[0],
                                                                           gflop / best[0] * 1e3, t2 / best[0]), flush=True)
