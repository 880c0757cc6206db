#!/usr/bin/env python
"""GPU microbench: the attention core of one FFT block (forward + backward) -- csrc/attn.hip against PyTorch-ROCm's fused
scaled_dot_product_attention on strided views of the same projection -- at the bench configuration's shapes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd')]
import msmctts_amd  # noqa
import torch
import torch.nn.functional as F
from msmctts_amd.hip import attn, norm

dev = torch.device('cuda:0')


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(3e7))
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for B, T, H, pd in ((16, 400, 2, 0.1), (16, 100, 2, 0.1), (16, 400, 2, 0.0), (4, 2400, 2, 0.1)):
    torch.manual_seed(0)
    qkv = (torch.randn(B, T, H * 192, device=dev) * 0.5).bfloat16().requires_grad_(True)
    pos = torch.arange(1, T + 1, device=dev).repeat(B, 1)
    pos[1:, T - T // 5:] = 0
    bias = attn.pad_key_bias(pos)
    add = torch.zeros(B, 1, 1, T, dtype=torch.bfloat16, device=dev).masked_fill_(pos.eq(0).view(B, 1, 1, T), float('-inf'))
    go = torch.randn(B, T, H * 64, device=dev).bfloat16()
    salt = norm.new_salt()

    def ours_f():
        return attn.attention(qkv, bias, H, 0.125, pd, salt)

    def ours_fb():
        qkv.grad = None
        ours_f().backward(go)

    def sdpa_f():
        x = qkv.view(B, T, H, 192).transpose(1, 2)
        o = F.scaled_dot_product_attention(x[..., :64], x[..., 64:128], x[..., 128:], attn_mask=add, dropout_p=pd, scale=0.125)
        return o.transpose(1, 2).reshape(B, T, H * 64)

    def sdpa_fb():
        qkv.grad = None
        sdpa_f().backward(go)

    with torch.no_grad():
        a, b = timed(ours_f), timed(sdpa_f)
    c, d = timed(ours_fb), timed(sdpa_fb)
    print('B %2d T %4d H %d p %.1f | forward: kernel %6.1f us, sdpa %6.1f us | forward+backward: kernel %6.1f us, sdpa %6.1f us'
          % (B, T, H, pd, a, b, c, d), flush=True)
