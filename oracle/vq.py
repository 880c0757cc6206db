"""Oracle: EMA vector quantiser (plain PyTorch fp32 + a numpy twin).  TEST INFRASTRUCTURE ONLY.

Restates reference msmctts/networks/vqgantts/modules.py
  * ``Quantize.forward``           :24-67   (search :26-31, gather :33, EMA :35-57, outputs :59-60)
  * ``MultiHeadQuantize.forward``  :137-151
Numerical contract: fp32 expanded distance
``|x|^2 - 2 x.E + |e|^2`` evaluated as three separately rounded terms, first-minimum tie rule,
EMA statistics over valid frames only, outputs formed with the pre-update codebook.
"""
import numpy as np
import torch
import torch.nn.functional as F


def head_distances(x2d, embed):
    """x2d (N, d), embed (d, K) -> (N, K) expanded squared distance (modules.py:26-30)."""
    return (x2d.pow(2).sum(1, keepdim=True) - 2 * x2d @ embed + embed.pow(2).sum(0, keepdim=True))


def quantize_head(x, length, embed, cluster_size, embed_avg, train_update, decay=0.99, eps=1e-5):
    """One head.  x (B, T, d); buffers are updated IN PLACE when ``train_update``.

    Returns (quantize_st, diff, ind) exactly as modules.py:59-67.
    """
    B, T, d = x.shape
    flat = x.reshape(-1, d)
    dist = head_distances(flat, embed)
    ind = (-dist).max(1)[1].view(B, T)
    q = F.embedding(ind, embed.transpose(0, 1))
    if train_update:
        with torch.no_grad():
            K = embed.shape[1]
            onehot = F.one_hot(ind, K).type(flat.dtype)
            keep = torch.arange(T, device=x.device)[None, :] < length.to(x.device)[:, None]
            onehot_v = onehot[keep]                 # same rows, same order as the reference's cat loop
            x_v = x.detach()[keep]
            count = onehot_v.sum(0)
            esum = x_v.transpose(0, 1) @ onehot_v
            cluster_size.mul_(decay).add_(count, alpha=1 - decay)
            embed_avg.mul_(decay).add_(esum, alpha=1 - decay)
            n = cluster_size.sum()
            smoothed = (cluster_size + eps) / (n + K * eps) * n
            embed.copy_(embed_avg / smoothed.unsqueeze(0))
    diff = (q.detach() - x).pow(2)
    q_st = x + (q - x).detach()
    return q_st, diff, ind


def multi_head_quantize(x, length, heads, train_update, decay=0.99, eps=1e-5):
    """heads: list of (embed, cluster_size, embed_avg).  modules.py:137-151.

    With a single head the reference instantiates ``Quantize`` directly (msmc_vqgan.py:127-128):
    indices are then (B, T) rather than (B, T, 1); callers handle that.
    """
    H = len(heads)
    chunks = torch.chunk(x, H, dim=-1)
    qs, ds, inds = [], [], []
    for xh, (e, c, a) in zip(chunks, heads):
        q, dff, ind = quantize_head(xh, length, e, c, a, train_update, decay, eps)
        qs.append(q)
        ds.append(dff)
        inds.append(ind)
    return torch.cat(qs, dim=-1), sum(ds) / H, torch.stack(inds, dim=-1)


# ----------------------------------------------------------------------------
# numpy twin for the large search-only cases used by the bench/roofline checks
# ----------------------------------------------------------------------------
def np_search(x, embeds):
    """x (N, D) float32, embeds list of H arrays (d, K) float32 -> (N, H) int64 first-min indices.

    fp32 arithmetic in the same three-term form; the GEMM accumulation order is BLAS's, so
    tests compare through the fp64 top-2-gap contract (tests/_vq_contract.py).
    """
    H = len(embeds)
    d = x.shape[1] // H
    out = np.empty((x.shape[0], H), dtype=np.int64)
    for h, e in enumerate(embeds):
        xh = x[:, h * d:(h + 1) * d]
        dist = (xh * xh).sum(1, keepdims=True, dtype=np.float32) - np.float32(2) * (xh @ e) \
            + (e * e).sum(0, keepdims=True, dtype=np.float32)
        out[:, h] = np.argmax(-dist, axis=1)
    return out
