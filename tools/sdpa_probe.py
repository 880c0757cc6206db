#!/usr/bin/env python
"""Is torch's fused scaled_dot_product_attention worth using for the FFT blocks' attention on this stack?
(B*H = 32, T = 400 / 100, d = 64, key-padding mask, dropout 0.1, bf16 autocast shapes; forward + backward)"""
import math, time, torch
import torch.nn.functional as F
dev = 'cuda:0'
torch.manual_seed(0)


def manual(q, k, v, mask, p):
    attn = torch.bmm(q, k.transpose(1, 2)) / 8.0
    attn = attn.masked_fill(mask, -math.inf)
    attn = torch.softmax(attn, dim=2)
    attn = F.dropout(attn, p, True)
    return torch.bmm(attn, v)


def fused(q, k, v, mask, p):
    return F.scaled_dot_product_attention(q.unsqueeze(0), k.unsqueeze(0), v.unsqueeze(0), attn_mask=~mask.unsqueeze(0),
                                          dropout_p=p).squeeze(0)


for T in (400, 100):
    for dt in (torch.bfloat16, torch.float32):
        q, k, v = [torch.randn(32, T, 64, device=dev, dtype=dt, requires_grad=True) for _ in range(3)]
        lens = torch.randint(T // 2, T + 1, (32,), device=dev)
        mask = (torch.arange(T, device=dev)[None, None, :] >= lens[:, None, None]).expand(32, T, T)
        with torch.no_grad():
            a, b = manual(q, k, v, mask, 0.0), fused(q, k, v, mask, 0.0)
        err = (a.float() - b.float()).abs().max().item()
        for name, fn in (('manual', manual), ('sdpa', fused)):
            for _ in range(3):
                fn(q, k, v, mask, 0.1).sum().backward()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn(q, k, v, mask, 0.1).sum().backward()
            torch.cuda.synchronize()
            print('T=%d %s %-6s fwd+bwd %.1f us   (max |manual - sdpa| without dropout: %.2e)'
                  % (T, str(dt)[6:], name, (time.perf_counter() - t0) / 20 * 1e6, err), flush=True)
