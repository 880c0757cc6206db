#!/usr/bin/env python
"""GPU microbench: the persistent thin-layer kernel (csrc/gather4.inc, descriptor variant 32) on the generator's C = 32 / 64
stages under different sizes of its persistent grid (include/msmc_hip_debug.h msmc_conv_set_gather4_grid; 0 = the product's one
or two workgroups per CU).  GRIDS="0 256 512 768 1024" selects the columns.

    python tools/bench_gather4_grid.py
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
L = lib.get()
SHAPES = [('rb C64 L6000 k3', 64, 6000, 3, 1), ('rb C64 L6000 k7', 64, 6000, 7, 1), ('rb C64 L6000 k11 d5', 64, 6000, 11, 5),
          ('rb C32 L12000 k3', 32, 12000, 3, 1), ('rb C32 L12000 k7 d3', 32, 12000, 7, 3), ('rb C32 L12000 k11', 32, 12000, 11, 1)]
GRIDS = [int(v) for v in os.environ.get('GRIDS', '0 256 384 512 768 1024 1536').split()]


def timed(desc, stream, iters=30):
    for _ in range(3):
        if L.msmc_conv_gather(ctypes.byref(desc), stream) != 0:
            return None
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(3e7))
    s.record()
    for _ in range(iters):
        L.msmc_conv_gather(ctypes.byref(desc), stream)
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print('%-24s | %s' % ('layer (B = 16, + residual)', ' '.join('g%-6d' % g for g in GRIDS)))
for name, C, Lp, k, dil in SHAPES:
    torch.manual_seed(0)
    geom = conv.Geometry(1, Lp, (1, k), (1, 1), (1, dil), (0, dil * (k - 1) // 2), False)
    x = torch.randn(16, 1, Lp, C, device=dev).bfloat16()
    w = (torch.randn(k, C, C, device=dev) / (C * k) ** 0.5).bfloat16()
    bias = torch.randn(C, device=dev)
    res = torch.randn(16, 1, Lp, C, device=dev).bfloat16()
    desc, out = conv._forward_desc(x, w, geom, bias, 0.1, res, None, 1.0, 1.0)
    desc.variant = 32
    cells = []
    for g in GRIDS:
        L.msmc_conv_set_gather4_grid(g)
        t = timed(desc, lib.stream(x))
        cells.append('%7.1f' % t if t else '    -  ')
    L.msmc_conv_set_gather4_grid(0)
    mb = (2 * x.numel() * 2 + x.numel() * 2 + w.numel() * 2) / 1e6
    print('%-24s | %s   (%.0f MB algorithmic)' % (name, ' '.join(cells), mb), flush=True)
