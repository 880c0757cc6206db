#!/usr/bin/env python
"""GPU diagnostic: weight/bias gradients of small shapes, FAST vs loop staging, against PyTorch-ROCm."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd'), os.path.join(ROOT, 'tests')]
import torch
import torch.nn.functional as F
from msmctts_amd.hip import conv, lib

DEV = 'cuda:0'
L = lib.get()
torch.manual_seed(0)
cases = [(3, 32, 32, 1, 8, (1, 7), (1, 1), (1, 1), (0, 3)), (3, 32, 32, 1, 48, (1, 3), (1, 1), (1, 1), (0, 1)),
         (3, 16, 16, 1, 240, (1, 11), (1, 1), (1, 5), (0, 25)), (3, 8, 8, 1, 1200, (1, 7), (1, 1), (1, 3), (0, 9)),
         (3, 4, 16, 400, 3, (5, 1), (3, 1), (1, 1), (2, 0)), (3, 32, 1, 1, 2400, (1, 7), (1, 1), (1, 1), (0, 3))]
for dt in (torch.float32, torch.bfloat16):
    for (B, Cin, Cout, H, W, k, s, dil, pad) in cases:
        x = torch.randn(B, Cin, H, W, device=DEV)
        w = torch.randn(Cout, Cin, *k, device=DEV, requires_grad=True)
        b = torch.randn(Cout, device=DEV, requires_grad=True)
        ref = F.conv2d(x, w, b, s, pad, dil)
        g = torch.randn_like(ref)
        ref.backward(g)
        geom = conv.Geometry(H, W, k, s, dil, pad)
        T = k[0] * k[1]
        xc = x.permute(0, 2, 3, 1).contiguous().to(dt)
        gc = g.permute(0, 2, 3, 1).contiguous().to(dt)
        want = w.grad.permute(2, 3, 0, 1).reshape(T, Cout, Cin)
        out = []
        for mode in (0, 1):
            L.msmc_conv_set_pipeline(mode)
            for rep in range(3):
                db = torch.zeros(Cout, device=DEV)
                dw = conv.conv_wgrad(xc, gc, geom, T, dw=torch.zeros(T, Cout, Cin, device=DEV), db=db)
                torch.cuda.synchronize()
                out.append(((dw - want).abs().max().item() / want.abs().max().item(),
                            (db - b.grad).abs().max().item() / b.grad.abs().max().item()))
        L.msmc_conv_set_pipeline(1)
        print(str(dt)[6:], (B, Cin, Cout, H, W, k), ' '.join('dw %.1e db %.1e |' % o for o in out), flush=True)
