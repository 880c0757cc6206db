#!/usr/bin/env python
"""GPU microbench: bf16 weight gradients of the bench configuration -- generation 2 (fp32 atomics, tuned split) against
generation 3 (split partials in a workspace + fixed-order second stage) and generation 4 (LDS-DMA ring of pixel tiles:
variants 4 / 5 / 6) and the general-lattice LDS-DMA form (variant 7, wgrad5.inc) at several split settings; results checked against generation 2.
CANDS="2/0 3/-1 4/0" selects the columns.

    python tools/bench_wgrad.py [filter]
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
flt = sys.argv[1] if len(sys.argv) > 1 else ''
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


cands = [(2, 0), (2, -1), (3, 0), (3, -1), (4, 0), (4, -1), (4, 1), (5, 0), (6, 0), (7, 0), (7, -1)]
if os.environ.get('CANDS'):
    cands = [tuple(int(v) for v in c.split('/')) for c in os.environ['CANDS'].split()]
print('%-28s %8s | %s' % ('layer', 'GFLOP', ' '.join('v%d/%+d   ' % c for c in cands)))
for name, B, Cin, Cout, H, W, k, s_, dil, pad, reflect, slope in SHAPES:
    if flt not in name:
        continue
    torch.manual_seed(0)
    geom = conv.Geometry(H, W, k, s_, dil, pad, reflect)
    T = k[0] * k[1]
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
    g = torch.randn(B, geom.Hout, geom.Wout, Cout, device=dev).bfloat16()
    copies = 8 if T * Cout * Cin <= 256 * 1024 else 1
    dw = torch.zeros(copies, T, Cout, Cin, device=dev)
    db = torch.zeros(copies, Cout, device=dev)
    desc = conv._build_desc(x.dtype, B, H, W, Cin, geom.Hout, geom.Wout, Cout, geom.fwd_lattice, geom.fwd_taps,
                            1 if reflect else 0, slope, 1.0, 1.0, 1.0)
    desc.x = desc.w = desc.out = x.data_ptr()
    desc.dw_copies = copies
    stream = lib.stream(x)
    ref = None
    cells, best = [], None
    for v, sh in cands:
        desc.variant, desc.split_shift = v, sh
        dw.zero_(); db.zero_()
        t = timed(desc, g.data_ptr(), dw, db, stream)
        if t is None:
            cells.append('   -    ')
            continue
        dw.zero_(); db.zero_()
        run(desc, g.data_ptr(), dw, db, stream)
        got = dw.sum(0)
        if ref is None:
            ref = got.clone()
        err = (got - ref).abs().max().item() / max(1e-6, ref.abs().max().item())
        cells.append('%7.1f%s' % (t, ' ' if err < 2e-3 else '!'))
        if best is None or t < best[0]:
            best = (t, v, sh)
    gflop = 2.0 * B * geom.Hout * geom.Wout * Cout * Cin * T / 1e9
    if best is None:
        best = (float('inf'), 0, 0)
    print('%-28s %8.2f | %s | best v%d/%+d %.1f us %.0f TF/s' % (name, gflop, ' '.join(cells), best[1], best[2], best[0],
                                                                gflop / best[0] * 1e-3), flush=True)
