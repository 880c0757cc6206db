#!/usr/bin/env python
"""GPU microbench: one bf16 weight gradient under forced pixel splits (msmc_conv_set_wgrad_split), per kernel variant --
separates the per-tile cost of the main loop (one split: every workgroup walks the whole pixel range, no second stage)
from the fixed costs (launch, partial stores, second stage).

    python tools/bench_wgrad_splits.py [filter]     VARIANTS="4 3" SPLITS="1 2 4 8 16"  ABLATE="0 1 2 3"
ABLATE (fourth generation only): 1 skips the MFMA steps, 2 the LDS-DMA stream, 3 both (barriers and waits remain).
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd'), os.path.join(ROOT, 'tools')]
import msmctts_amd  # noqa
import torch
from msmctts_amd.hip import conv, lib
sys.argv, _argv = sys.argv[:1] + ['__no_such_layer__'], sys.argv
from bench_gather3 import SHAPES
sys.argv = _argv

dev = torch.device('cuda:0')
flt = sys.argv[1] if len(sys.argv) > 1 else 'ffn w1 T400'
L = lib.get()


def run(desc, gp, dw, db, stream):
    need = L.msmc_conv_wgrad_workspace(ctypes.byref(desc), gp)
    wsp, wsb = conv._workspace(dw.device, stream, need) if need else (None, 0)
    return L.msmc_conv_wgrad_ws(ctypes.byref(desc), gp, dw.data_ptr(), db.data_ptr(), wsp, wsb, stream)


def timed(desc, gp, dw, db, stream, iters=10):
    if run(desc, gp, dw, db, stream) != 0:
        return None
    run(desc, gp, dw, db, stream)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(3e7))
    s.record()
    for _ in range(iters):
        run(desc, gp, dw, db, stream)
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters * 1e3


variants = [int(v) for v in os.environ.get('VARIANTS', '4 3').split()]
ablates = [int(v) for v in os.environ.get('ABLATE', '0').split()]
splits = [int(v) for v in os.environ.get('SPLITS', '1 2 4 8 16 32').split()]
print('%-28s %s' % ('layer / variant', ' '.join('n=%-6d' % n for n in splits)))
for name, B, Cin, Cout, H, W, k, s_, dil, pad, reflect, slope in SHAPES:
    if flt not in name:
        continue
    geom = conv.Geometry(H, W, k, s_, dil, pad, reflect)
    T = k[0] * k[1]
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
    g = torch.randn(B, geom.Hout, geom.Wout, Cout, device=dev).bfloat16()
    dw, db = torch.zeros(1, T, Cout, Cin, device=dev), torch.zeros(1, Cout, device=dev)
    desc = conv._build_desc(x.dtype, B, H, W, Cin, geom.Hout, geom.Wout, Cout, geom.fwd_lattice, geom.fwd_taps,
                            1 if reflect else 0, slope, 1.0, 1.0, 1.0)
    desc.x = desc.w = desc.out = x.data_ptr()
    desc.dw_copies = 1
    for v, abl in [(v, a) for v in variants for a in ablates if a == 0 or v >= 4]:
        desc.variant, desc.split_shift = v, 0
        L.msmc_conv_set_wgrad4_ablate(abl)
        cells = []
        for n in splits:
            L.msmc_conv_set_wgrad_split(n)
            t = timed(desc, g.data_ptr(), dw, db, lib.stream(x))
            cells.append('%7.1f ' % t if t is not None else '   -    ')
        L.msmc_conv_set_wgrad_split(0)
        L.msmc_conv_set_wgrad4_ablate(0)
        print('%-28s %s' % ('%s v%d%s' % (name[:20], v, ' a%d' % abl if abl else ''), ' '.join(cells)), flush=True)
