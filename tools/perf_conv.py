#!/usr/bin/env python
"""Per-layer timing of the implicit-GEMM convolution kernels at the CSMSC shapes (B=16, bf16) on the GPU:
forward / data-gradient / weight-gradient, simple vs pipelined gather kernel, next to PyTorch-ROCm (MIOpen)
for the same layer.  Prints a table and writes gpurun_out/perf_conv.json.   Usage: python tools/perf_conv.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd'), os.path.join(ROOT, 'tests')]
import torch
import torch.nn.functional as F

from msmctts_amd.hip import conv, lib
from test_gpu_conv import CONVS

DEV = 'cuda:0'


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3       # us


def main():
    dt = torch.bfloat16
    rows = []
    L = lib.get()
    for name, B, Cin, Cout, H, W, k, s, dil, pad, reflect, slope in CONVS:
        B = 16
        geom = conv.Geometry(H, W, k, s, dil, pad, reflect)
        T = k[0] * k[1]
        x = torch.randn(B, H, W, Cin, device=DEV).to(dt)
        g = torch.randn(B, geom.Hout, geom.Wout, Cout, device=DEV).to(dt)
        wf = (torch.randn(T, Cout, Cin, device=DEV) * 0.05).to(dt)
        wb = (torch.randn(T, Cin, Cout, device=DEV) * 0.05).to(dt)
        bias = torch.randn(Cout, device=DEV)
        dw = torch.zeros(T, Cout, Cin, device=DEV)
        db = torch.zeros(Cout, device=DEV)
        flops = 2.0 * B * geom.Hout * geom.Wout * Cout * Cin * T
        r = {'layer': name, 'gflop': flops / 1e9}
        for mode, tag in ((0, 'simple'), (1, 'pipe')):
            L.msmc_conv_set_pipeline(mode)
            r['fwd_' + tag] = timeit(lambda: conv.conv_forward(x, wf, geom, bias=bias, in_slope=slope))
            r['dgrad_' + tag] = timeit(lambda: conv.conv_dgrad(g, wb, geom))
        L.msmc_conv_set_pipeline(1)
        L.msmc_conv_set_narrow(0)
        r['fwd_wide'] = timeit(lambda: conv.conv_forward(x, wf, geom, bias=bias, in_slope=slope))
        L.msmc_conv_set_narrow(1)
        r['wgrad'] = timeit(lambda: conv.conv_wgrad(x, g, geom, T, in_slope=slope, dw=dw, db=db))
        r['wgrad_sweep'] = {}
        for ns in (1, 2, 4, 8, 16, 32, 64, 128, 512):
            L.msmc_conv_set_wgrad_split(ns)
            r['wgrad_sweep'][ns] = timeit(lambda: conv.conv_wgrad(x, g, geom, T, in_slope=slope, dw=dw, db=db), iters=10)
        L.msmc_conv_set_wgrad_split(0)
        # PyTorch-ROCm reference timing (NCHW bf16, MIOpen)
        xn = x.permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        wn = wf.permute(1, 2, 0).reshape(Cout, Cin, *k).contiguous().requires_grad_(True)
        gn = g.permute(0, 3, 1, 2).contiguous()
        padt = (pad[1], pad[1], pad[0], pad[0])

        def torch_fwd():
            xi = F.pad(xn, padt, mode='reflect') if reflect else xn
            return F.conv2d(xi, wn, bias.to(dt), s, 0 if reflect else pad, dil)
        try:
            r['torch_fwd'] = timeit(torch_fwd)
            out = torch_fwd()
            r['torch_bwd'] = timeit(lambda: torch.autograd.grad(out, (xn, wn), gn, retain_graph=True))
        except Exception as ex:          # noqa
            r['torch_fwd'] = r['torch_bwd'] = float('nan')
        rows.append(r)
        print('%-28s %7.2f GF | fwd %7.1f -> %7.1f us (wide %7.1f) | dgrad %7.1f -> %7.1f | wgrad %7.1f | '
              'torch fwd %7.1f bwd %7.1f | wgrad split sweep %s' % (
                  name, r['gflop'], r['fwd_simple'], r['fwd_pipe'], r['fwd_wide'], r['dgrad_simple'], r['dgrad_pipe'],
                  r['wgrad'], r['torch_fwd'], r['torch_bwd'],
                  ' '.join('%d:%.0f' % kv for kv in r['wgrad_sweep'].items())), flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'perf_conv.json'), 'w') as f:
        json.dump(rows, f, indent=1)
    tot = lambda key: sum(r[key] for r in rows)
    print('SUM us: fwd simple %.0f pipe %.0f | dgrad simple %.0f pipe %.0f | wgrad %.0f | torch fwd %.0f bwd %.0f'
          % (tot('fwd_simple'), tot('fwd_pipe'), tot('dgrad_simple'), tot('dgrad_pipe'), tot('wgrad'), tot('torch_fwd'),
             tot('torch_bwd')))


if __name__ == '__main__':
    main()
