"""EMA vector quantisers and the WaveNet residual stack (drop-in for reference
msmctts/networks/vqgantts/modules.py).

``Quantize`` / ``MultiHeadQuantize`` keep the reference's constructor arguments, buffer names and
layouts (``embed`` (sub_dim, K), ``cluster_size`` (K,), ``embed_avg`` (sub_dim, K)) and call
signatures, but one forward is three HIP launches for *all* heads (csrc/vq.hip): codebook
transpose + norms, fused search/gather/straight-through/squared-error, and -- in training with
``update`` -- the deterministic EMA update.  The per-head buffers are views into head-packed
tensors so the kernels see [H][d][K] while ``state_dict`` keeps the reference keys.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...hip import losses as hiploss
from ...hip import vq as hipvq
from ..layers import WNConv1d


def _ranking(input, embed):
    """``sort=True`` of the reference (modules.py:62-65): ALL codewords of every frame ordered by distance, nearest
    first -- [..., K] per head.  No training or inference path of the reference asks for it; stock operators on the
    expanded distance matrix, evaluated with the codebook the search used (before the EMA update)."""
    H, d, K = embed.shape
    flat = input.detach().reshape(-1, H, d).float()
    dist = (flat.pow(2).sum(-1, keepdim=True) - 2 * torch.einsum('nhd,hdk->nhk', flat, embed)
            + embed.pow(2).sum(1).unsqueeze(0))
    order = (-dist).sort(dim=-1, descending=True)[1]                       # [N, H, K]
    return order.reshape(*input.shape[:-1], H, K)


def _ema(q, x3, ind, length, embed, cs, ea):
    """EMA codebook update of one stage: fused (reference semantics: this rank's frames only) or, with
    ``q.sync_stats`` (VQGANTrainer sync_codebook_stats=True), statistics now and the update after the cross-rank sum
    (hip/vq.py flush_codebook_sync, called by the trainer between forward and backward)."""
    if not getattr(q, 'sync_stats', False):
        q._ws = hipvq.vq_ema_update(x3, ind, length, embed, cs, ea, q.decay, q.eps, q._ws)
        return
    cb = getattr(q, '_sync', None)
    if cb is None or not cb.matches(embed):
        cb = q._sync = hipvq.CodebookSync(embed, cs, ea, q.decay, q.eps)
    cb.collect(x3, ind, length)
    if not any(c is cb for c in hipvq.PENDING):
        hipvq.PENDING.append(cb)


class Quantize(nn.Module):
    """Single-codebook EMA quantiser (reference modules.py:10-116)."""

    def __init__(self, embed_dim, n_embed, decay=0.99, eps=1e-5):
        super().__init__()
        self.dim, self.n_embed, self.decay, self.eps = embed_dim, n_embed, decay, eps
        embed = torch.randn(embed_dim, n_embed)
        self.register_buffer('embed', embed)
        self.register_buffer('cluster_size', torch.zeros(n_embed))
        self.register_buffer('embed_avg', embed.clone())
        self._ws = None

    def _packed(self):
        return self.embed.unsqueeze(0), self.cluster_size.unsqueeze(0), self.embed_avg.unsqueeze(0)

    def forward(self, input, input_length=None, update=True, sort=False):
        embed, cs, ea = self._packed()
        embed_t, enorm = hipvq.vq_prepare(embed, frames=input.numel() // max(1, input.shape[-1]))
        quant, diff, ind = hipvq.vq_search(input, embed_t, enorm)
        rank = _ranking(input, embed).squeeze(-2) if sort else None         # [..., K]
        if self.training and update and input.numel() > 0:          # (an empty batch has no statistics to add)
            x3 = input.detach().reshape(input.shape[0], -1, input.shape[-1])
            _ema(self, x3, ind.reshape(x3.shape[0], x3.shape[1], -1), input_length, embed, cs, ea)
        return quant, diff, rank if sort else ind.squeeze(-1)

    def embed_code(self, embed_id):
        return F.embedding(embed_id, self.embed.transpose(0, 1))

    def compute_triple_loss(self, prd_quant, trg_quant, reduction='mean', margin=1e-6, adaptive_margin=False):
        """hinge of (squared distance to the target codeword) against the squared distance to every codeword (reference
        modules.py:86-116, the 'masked version'): [B, T] per-frame loss.  On the GPU one launch per call (round 5); the stock
        operator chain on the expanded distance matrix below serves reduction='none' and odd codeword widths."""
        B, T, D = prd_quant.shape
        if (reduction in ('mean', 'sum') and hiploss.usable(prd_quant) and self.dim in (16, 32, 64, 128)
                and (self.n_embed * self.dim + self.n_embed) * 4 <= 160 * 1024):       # (the head's codebook sits in LDS)
            # one launch: all K distances, the hinge and its gradient per frame (csrc/losses.hip triple_loss_kernel)
            embed_t, enorm = hipvq.vq_prepare(self.embed.unsqueeze(0).contiguous(), frames=0)
            return hiploss.triple_loss(prd_quant.reshape(-1, D), trg_quant.reshape(-1, 1), embed_t, enorm, reduction,
                                       margin).reshape(B, T)
        flat = prd_quant.reshape(-1, self.dim)
        dist = (flat.pow(2).sum(1, keepdim=True) - 2 * flat @ self.embed + self.embed.pow(2).sum(0, keepdim=True)).reshape(B, T, -1)
        pos = F.mse_loss(prd_quant, self.embed_code(trg_quant), reduction='none').sum(-1)
        triple = pos.unsqueeze(-1) - dist
        mask = triple != 0
        triple = mask * (torch.clamp(triple + margin, min=0) / self.dim)
        return triple.mean(-1) if reduction == 'mean' else triple.sum(-1) if reduction == 'sum' else triple


class MultiHeadQuantize(nn.Module):
    """H independent codebooks over contiguous feature chunks (reference modules.py:119-169)."""

    def __init__(self, embed_dim, n_embed, n_head, decay=0.99, eps=1e-5):
        super().__init__()
        assert embed_dim % n_head == 0
        self.dim, self.n_embed, self.n_head, self.decay, self.eps = embed_dim, n_embed, n_head, decay, eps
        self.quantizers = nn.ModuleList([Quantize(embed_dim // n_head, n_embed, decay, eps) for _ in range(n_head)])
        self._pack = None
        self._ws = None

    def _packed(self):
        """Head-packed buffers; (re)built whenever the per-head buffers stopped being views of them
        (construction, .to()/.cuda(), a non-in-place load)."""
        qs = self.quantizers
        p = self._pack
        ok = p is not None and all(
            q.embed.data_ptr() == p[0][h].data_ptr() and q.cluster_size.data_ptr() == p[1][h].data_ptr()
            and q.embed_avg.data_ptr() == p[2][h].data_ptr() for h, q in enumerate(qs))
        if not ok:
            p = (torch.stack([q.embed for q in qs]).contiguous(), torch.stack([q.cluster_size for q in qs]).contiguous(),
                 torch.stack([q.embed_avg for q in qs]).contiguous())
            for h, q in enumerate(qs):
                q._buffers['embed'], q._buffers['cluster_size'], q._buffers['embed_avg'] = p[0][h], p[1][h], p[2][h]
            self._pack = p
        return p

    def forward(self, input, input_length=None, update=True, sort=False):
        embed, cs, ea = self._packed()
        embed_t, enorm = hipvq.vq_prepare(embed, frames=input.numel() // max(1, input.shape[-1]))
        quant, diff, ind = hipvq.vq_search(input, embed_t, enorm)
        rank = _ranking(input, embed).transpose(-1, -2) if sort else None   # [..., K, H], as the reference's stack
        if self.training and update and input.numel() > 0:          # (an empty batch has no statistics to add)
            x3 = input.detach().reshape(input.shape[0], -1, input.shape[-1])
            _ema(self, x3, ind.reshape(x3.shape[0], x3.shape[1], -1), input_length, embed, cs, ea)
        return quant, diff, rank if sort else ind


    def compute_triple_loss(self, prd_quant, trg_quant, reduction='mean', margin=1e-6, adaptive_margin=False):
        """mean over heads of the per-head triple loss (reference modules.py:152-168); trg_quant [B, T, H] indices"""
        d = self.dim // self.n_head
        if (reduction in ('mean', 'sum') and hiploss.usable(prd_quant) and d in (16, 32, 64, 128)
                and (self.n_embed * d + self.n_embed) * 4 <= 160 * 1024 and trg_quant.shape[-1] == self.n_head):
            # all heads in one launch on the packed codebook; the mean over heads as the reference's sum(losses) / len(losses)
            B, T, D = prd_quant.shape
            embed, _, _ = self._packed()
            embed_t, enorm = hipvq.vq_prepare(embed, frames=0)            # (no shortlist image: only rows and squared norms)
            lossh = hiploss.triple_loss(prd_quant.reshape(-1, D), trg_quant.reshape(-1, self.n_head), embed_t, enorm, reduction,
                                        margin)
            return (lossh.sum(-1) / self.n_head).reshape(B, T)
        prds = torch.chunk(prd_quant, self.n_head, dim=-1)
        trgs = torch.chunk(trg_quant, self.n_head, dim=-1)
        losses = []
        for q, p, t in zip(self.quantizers, prds, trgs):
            assert t.shape[-1] == 1
            losses.append(q.compute_triple_loss(p, t.squeeze(-1), reduction, margin, adaptive_margin))
        return sum(losses) / len(losses)


class ResStack(nn.Module):
    """Un-conditioned WaveNet stack used by PriorPredictor (reference modules.py:182-259).

    The reference materialises an all-zero conditioning tensor per layer (modules.py:236); adding
    zeros is the identity, so it is skipped.  ``p_dropout`` stays hard-wired at 0.1 like the reference.
    """

    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0.1):
        super().__init__()
        assert kernel_size % 2 == 1
        assert gin_channels == 0, 'conditioning is never used on the MSMC-VQ-GAN path'
        self.hidden_channels, self.n_layers = hidden_channels, n_layers
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        self.drop = nn.Dropout(p_dropout)
        for i in range(n_layers):
            dil = dilation_rate ** i
            self.in_layers.append(WNConv1d(hidden_channels, 2 * hidden_channels, kernel_size, dilation=dil,
                                           padding=int((kernel_size * dil - dil) / 2)))
            out_ch = 2 * hidden_channels if i < n_layers - 1 else hidden_channels
            self.res_skip_layers.append(WNConv1d(hidden_channels, out_ch, 1))

    def forward(self, x, x_mask, g=None, **kwargs):
        C = self.hidden_channels
        output = None
        for i in range(self.n_layers):
            x_in = self.in_layers[i](x)
            acts = self.drop(torch.tanh(x_in[:, :C]) * torch.sigmoid(x_in[:, C:]))
            rs = self.res_skip_layers[i](acts)
            if i < self.n_layers - 1:
                x = (x + rs[:, :C]) * x_mask
                skip = rs[:, C:]
            else:
                skip = rs
            output = skip if output is None else output + skip
        return output * x_mask
