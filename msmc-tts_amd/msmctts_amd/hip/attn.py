"""Attention core of the FFT blocks on the gfx950 kernels (csrc/attn.hip): autograd function over the fused projection.

``attention(qkv, key_bias, n_head, scale, p_drop, salt)``: qkv [B, T, H*192] bf16 (per head q | k | v, 64 each) ->
[B, T, H*64]; ``key_bias`` [B, Tp] fp32 from ``pad_key_bias`` (0 = attend, -inf = padding).  The backward pass recomputes
the probabilities from the saved log-sum-exp and regenerates the dropout mask from the same (seed word, salt)."""
import torch

from . import lib
from .norm import seed_word

HEAD = 64


def supported(dtype, d_k, d_v):
    return dtype == torch.bfloat16 and d_k == HEAD and d_v == HEAD


def pad_key_bias(pos):
    """pos [B, T] (0 = padding) -> additive key bias [B, Tp], Tp = T rounded up to 32, -inf on padding and on the tail"""
    B, T = pos.shape
    Tp = (T + 31) // 32 * 32
    bias = torch.full((B, Tp), float('-inf'), dtype=torch.float32, device=pos.device)
    bias[:, :T].masked_fill_(pos.ne(0), 0.0)
    return bias


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, bias, H, scale, p_drop, salt):
        B, T, E = qkv.shape
        assert E == H * 3 * HEAD and qkv.dtype == torch.bfloat16 and qkv.is_contiguous() and bias.dtype == torch.float32
        out = torch.empty(B, T, H * HEAD, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(B * H, T, dtype=torch.float32, device=qkv.device)
        seed = seed_word(qkv.device) if p_drop > 0 else None
        lib.check(lib.get().msmc_attn_fwd(lib.ptr(qkv), lib.ptr(bias), lib.ptr(out), lib.ptr(lse), B, T, H, bias.shape[1],
                                          float(scale), float(p_drop), lib.ptr(seed) if seed is not None else None, int(salt),
                                          lib.stream(qkv)), 'msmc_attn_fwd')
        ctx.save_for_backward(qkv, bias, out, lse)
        ctx.args = (H, float(scale), float(p_drop), int(salt))
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, bias, out, lse = ctx.saved_tensors
        H, scale, p_drop, salt = ctx.args
        B, T, _ = qkv.shape
        g = g.contiguous()
        if g.dtype != qkv.dtype:
            g = g.to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        dsum = torch.empty(B * H, T, dtype=torch.float32, device=qkv.device)
        seed = seed_word(qkv.device) if p_drop > 0 else None
        lib.check(lib.get().msmc_attn_bwd(lib.ptr(qkv), lib.ptr(bias), lib.ptr(out), lib.ptr(lse), lib.ptr(g), lib.ptr(dqkv),
                                          lib.ptr(dsum), B, T, H, bias.shape[1], scale, p_drop,
                                          lib.ptr(seed) if seed is not None else None, salt, lib.stream(qkv)), 'msmc_attn_bwd')
        return dqkv, None, None, None, None, None


def attention(qkv, key_bias, n_head, scale, p_drop=0.0, salt=0):
    return _Attention.apply(qkv, key_bias, int(n_head), float(scale), float(p_drop), int(salt))
