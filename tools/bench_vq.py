#!/usr/bin/env python
"""Runs ON the GPU box: VQ search kernels side by side -- the bf16-shortlist kernel (csrc/vq_shortlist.inc) against the exact
register-resident kernel -- over frame counts from the training step's (1 600 / 6 400) to the micro-benchmark's 2^20, with the
fractions of 16-frame tiles that left the shortlist path.  Results are compared bit for bit on the way."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd')]
import msmctts_amd  # noqa
import torch
from msmctts_amd.hip import lib, vq

dev = torch.device('cuda:0')
ABLATE = int(os.environ.get('ABLATE', '0'))      # msmc_vq_set_shortlist_ablate mask (timings only; results are garbage)
lib.get().msmc_vq_set_shortlist_ablate(ABLATE)
print('ablate mask', ABLATE)
NS = [int(v) for v in os.environ.get('NS', '1600,6400,25600,131072,1048576').split(',')]
CFG = [(4, 64), (4, 256), (8, 512)]
print('%-10s %9s | %-22s %9s %8s | %-22s %9s %8s | same  rerank   research' % ('H x K', 'N', 'product', 'us', 'GB/s', 'exact', 'us', 'GB/s'))
for H, K in CFG:
    D = 256
    g = torch.Generator().manual_seed(0)
    e = torch.randn(H, D // H, K, generator=g).to(dev)
    et, en = vq.vq_prepare(e)
    for N in NS:
        x = torch.randn(N, D, generator=g).to(dev)
        byts = N * (8 * D + 8 * H + 4 * D // H)
        res = []
        for sl in (None, False):
            for _ in range(3):
                out = vq.vq_search(x, et, en, shortlist=sl)
            torch.cuda.synchronize()
            iters = 20 if N <= 200000 else 10
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                out = vq.vq_search(x, et, en, shortlist=sl)
            t.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(t) / iters * 1e3
            # the exact-path counters in a run of their own: tens of thousands of atomics on one address would be the
            # slowest thing in the timed launches
            vq.SLOW_COUNT = torch.zeros(2, dtype=torch.int64, device=dev)
            vq.vq_search(x, et, en, shortlist=sl)
            torch.cuda.synchronize()
            res.append((lib.get().msmc_vq_last_kernel().decode(), us, byts / us / 1e3, out, [float(v) for v in vq.SLOW_COUNT.tolist()]))
            vq.SLOW_COUNT = None
        if ABLATE & 32:                                  # phase timing of the waves of workgroup 0 (see vq_shortlist.inc)
            vq.SLOW_COUNT = torch.zeros(2 + 8 * 16, dtype=torch.int64, device=dev)
            vq.vq_search(x, et, en)
            torch.cuda.synchronize()
            c = vq.SLOW_COUNT.tolist()
            vq.SLOW_COUNT = None
            print('   cycles per step, waves of workgroup 0: barrier+dma | search loop | decide | gather+wait+convert | epilogue+stores | frame requests | total')
            for wv in range(16):
                r = c[2 + 8 * wv:2 + 8 * wv + 7]
                if r[6]:
                    print('     wave %2d (%3d steps): %s | %.0f' % (wv, r[6], ' | '.join('%6.0f' % (v / r[6]) for v in r[:6]), sum(r[:6]) / r[6]))
        same = all(torch.equal(a, b) for a, b in zip(res[0][3], res[1][3]))
        tiles = (N + 15) // 16 * H
        print('%-10s %9d | %-22s %9.1f %8.0f | %-22s %9.1f %8.0f | %s  %.4f  %.5f' % (
            '%dx%d' % (H, K), N, res[0][0], res[0][1], res[0][2], res[1][0], res[1][1], res[1][2], same,
            res[0][4][0] / tiles, res[0][4][1] / tiles))
        sys.stdout.flush()
