#!/usr/bin/env python
"""GPU probe: where a seventh-generation forward-convolution launch spends its time.  Needs a probe build of the library
(csrc/conv.hip compiled with -DCV7_CLOCKS, linked as lib/libmsmc_hip_clk.so: tools/r05_clock_probe.sh): with bit 64 of the
diagnostics mask the kernel stamps the shader clock at its phase boundaries per workgroup.  Prints, per (shape, variant):
the distribution over the workgroups of the phases (entry -> first requests issued, first data landed, main loop, exchange of
the contraction groups, epilogue incl. store drain) as min / median / max shader cycles.

    MSMC_PROBE_LIB=msmc-tts_amd/lib/libmsmc_hip_clk.so python tools/clock_probe.py [filter]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd')]
import msmctts_amd  # noqa
import torch
from msmctts_amd.hip import conv, lib

lib._lib = lib.load(os.environ.get('MSMC_PROBE_LIB', os.path.join(ROOT, 'msmc-tts_amd', 'lib', 'libmsmc_hip_clk.so')))
L = lib.get()
dev = torch.device('cuda:0')
SHAPES = [
    ('ffn w1 T400 256->1024 k3', 16, 256, 1024, 1, 400, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0),
    ('ffn w2 T400 1024->256 k3', 16, 1024, 256, 1, 400, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0),
    ('ffn w1 T100 256->1024 k3', 16, 256, 1024, 1, 100, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0),
    ('rb C128 L1200 k7 d3', 16, 128, 128, 1, 1200, (1, 7), (1, 1), (1, 3), (0, 9), False, 0.1),
    ('mpd p2 512->512 s1', 32, 512, 512, 75, 2, (5, 1), (1, 1), (1, 1), (2, 0), False, 0.2),
]
flt = sys.argv[1] if len(sys.argv) > 1 else ''
VARIANTS = [int(v) for v in os.environ.get('VARIANTS', '56 57 58 60 63').split()]
ABLS = [int(v) for v in os.environ.get('ABLS', '0 46').split()]
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
    desc, out = conv._forward_desc(x, w, geom, bias, slope, res, None, 1.0, 1.0)
    stream = lib.stream(x)
    for v in VARIANTS:
        for abl in ABLS:
            d = lib.ConvDesc.from_buffer_copy(desc)
            stamps = torch.zeros(8192, 8, dtype=torch.int64, device=dev)
            d.variant, d.split_shift, d.res2 = v, abl | 64, stamps.data_ptr()
            rc = L.msmc_conv_gather(ctypes.byref(d), stream)
            if rc != 0:
                continue
            for _ in range(3):                      # warm: the last launch's stamps are the ones read
                stamps.zero_()
                torch.cuda.synchronize()
                L.msmc_conv_gather(ctypes.byref(d), stream)
            torch.cuda.synchronize()
            st = stamps.cpu()
            st = st[st[:, 0] != 0]
            # (the shader clock is per XCD: only differences INSIDE a workgroup are meaningful -- no launch-wide total, no start skew)
            rel = (st[:, :6] - st[:, :1]).double()
            q = lambda t: '%6.0f/%6.0f/%6.0f' % (t.min().item(), t.median().item(), t.max().item())
            print('%-26s v%d abl%-3d workgroups %4d | requests issued %s | first data %s | loop %s | exchange %s | epilogue + store drain %s | '
                  'workgroup total %s' % (
                      name, v, abl, st.shape[0], q(rel[:, 1] - rel[:, 0]), q(rel[:, 2] - rel[:, 1]), q(rel[:, 3] - rel[:, 2]),
                      q(rel[:, 4] - rel[:, 3]), q(rel[:, 5] - rel[:, 4]), q(rel[:, 5])), flush=True)
