#!/usr/bin/env python
"""Runs ON the GPU box: the fused dropout + residual + LayerNorm kernels (csrc/norm.hip) forward and backward at the FFT-block shapes of
the bench configurations -- 6 400 x 256 (configuration 2) and 25 600 x 600 (configuration 4) -- per-launch time from the library's
own event pairs and the algorithmic bytes / time of each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd')]
import msmctts_amd  # noqa
import torch
from msmctts_amd.hip import norm

dev = torch.device('cuda:0')
for N, C, P in ((6400, 256, 0.1), (25600, 600, 0.1), (25600, 1024, 0.1), (25600, 608, 0.0)):
    x = torch.randn(N, C, device=dev).bfloat16().requires_grad_(True)
    r = torch.randn(N, C, device=dev).bfloat16().requires_grad_(True)
    gamma, beta = torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True)
    keep = torch.ones(N, dtype=torch.uint8, device=dev)
    g = torch.randn(N, C, device=dev).bfloat16()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    iters = 20
    for it in range(iters + 3):
        ev[0].record()
        y = norm.add_layer_norm(x, r, gamma, beta, keep, P, 7)
        ev[1].record()
        y.backward(g)
        ev[2].record()
        torch.cuda.synchronize()
        if it >= 3:
            tf += ev[0].elapsed_time(ev[1]) * 1e3 / iters
            tb += ev[1].elapsed_time(ev[2]) * 1e3 / iters
    byts = N * C * 2 * 4
    print('%6d x %4d bf16: forward %7.1f us (%5.0f GB/s), backward incl. parameter reduction and host %7.1f us (%5.0f GB/s)' % (N, C, tf, byts / tf / 1e3, tb, byts / tb / 1e3), flush=True)
