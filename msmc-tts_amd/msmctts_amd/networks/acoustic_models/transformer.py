"""Feed-forward Transformer (FFT) blocks used by the multi-stage encoder and the frame decoder
(drop-in for reference msmctts/networks/acoustic_models/transformer.py:71-424).

Module / parameter names and numerics follow the reference: fused QKV projection, post-LayerNorm, key-padding mask
with -inf, conv feed-forward, output zeroed on padding after both sub-layers.  On the GPU a block is eight launches
forward: the fused-QKV and output projections are 1-tap convolutions on the gfx950 implicit-GEMM
kernels (bias fused, activations stay in the compute dtype: no cast kernels), the position-wise convolutions run on
the same kernels, and ``layer_norm(dropout(h) + residual) * non_pad_mask`` is one fused kernel per sub-layer
(csrc/norm.hip, masks regenerated in the backward pass).  The softmax-attention core runs on csrc/attn.hip (bf16, head size 64:
the benchmarked configuration); fp32 runs -- the parity configuration -- and other head sizes use PyTorch-ROCm's fused
``scaled_dot_product_attention`` on strided views of the QKV projection.  There is no stock-operator version of the block
stack: on a device without the library (or the test interpreter) the forward raises.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...hip import attn as hipattn
from ...hip import norm as hipnorm
from ...hip.convnet import ConvBank, ConvLayer, hip_conv, hip_conv_add_ln


def get_sinusoid_encoding_table(n_position, d_hid, padding_idx=None):
    """(n_position, d_hid) float32 table, row ``padding_idx`` zeroed (transformer.py:388-407)."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    idx = np.arange(d_hid)[None, :]
    angle = pos / np.power(10000, 2 * (idx // 2) / d_hid)
    table = np.empty_like(angle)
    table[:, 0::2] = np.sin(angle[:, 0::2])
    table[:, 1::2] = np.cos(angle[:, 1::2])
    if padding_idx is not None:
        table[padding_idx] = 0.0
    return torch.FloatTensor(table)


def get_attn_key_pad_mask(seq_k, seq_q):
    return seq_k.eq(0).unsqueeze(1).expand(-1, seq_q.size(1), -1)


def get_non_pad_mask(seq):
    assert seq.dim() == 2
    return seq.ne(0).unsqueeze(-1)


# 1 (default): the head of FFTBlocks.forward as one launch when the caller passes ``lengths`` (20.59 -> 20.41 ms/step, round 4);
# MSMC_FFT_PROLOGUE=0 keeps the operator chain (A/B)
FFT_PROLOGUE = os.environ.get('MSMC_FFT_PROLOGUE', '1') == '1'


class ScaledDotProductAttention(nn.Module):
    """temperature and dropout probability of the attention core (the core itself: MultiHeadAttention.forward_hip)"""

    def __init__(self, temperature, attn_dropout=0.1, name=None):
        super().__init__()
        self.temperature, self.name = temperature, name
        if attn_dropout > 0:
            self.dropout = nn.Dropout(attn_dropout)


class _SplitHeads(torch.autograd.Function):
    """q, k, v as strided views of the fused projection [bs, H, T, 2*dk + dv]; the backward pass writes the three
    gradients side by side in ONE concatenation (slicing's own backward zero-fills and copies a full-size tensor per
    slice: six launches per attention layer)."""

    @staticmethod
    def forward(ctx, qkv, dk):
        ctx.dk = dk
        return qkv[..., :dk], qkv[..., dk:2 * dk], qkv[..., 2 * dk:]

    @staticmethod
    def backward(ctx, gq, gk, gv):
        # assembled in the memory order of the projection's output ([bs, T, H, E]) and handed back as the transposed view,
        # so the projection's backward reads it without a layout copy
        return torch.cat((gq.transpose(1, 2), gk.transpose(1, 2), gv.transpose(1, 2)), dim=-1).transpose(1, 2), None


class MultiHeadAttention(nn.Module):
    def __init__(self, n_head, d_model, d_k, d_v, dropout, name, attn_dropout=0.1, fused_layernorm=False):
        super().__init__()
        self.n_head, self.d_k, self.d_v, self.name = n_head, d_k, d_v, name
        self.linear = nn.Linear(d_model, n_head * (2 * d_k + d_v))
        nn.init.xavier_normal_(self.linear.weight)
        self.attention = ScaledDotProductAttention(float(np.power(d_k, 0.5)), attn_dropout, name + '.scaled_dot')
        self.layer_norm = nn.LayerNorm(d_model)      # apex FusedLayerNorm is disabled in every shipped config
        self.fc = nn.Linear(n_head * d_v, d_model)
        nn.init.xavier_normal_(self.fc.weight)
        self.dropout = nn.Dropout(dropout)
        self._salt = hipnorm.new_salt()
        self._salt_attn = hipnorm.new_salt()

    def hip_layers(self):
        """the two projections as 1-tap convolutions of the block stack's ConvBank (a Linear weight (out, in) is the
        (out, in, 1) weight of a kernel-size-1 convolution)"""
        return [ConvLayer(m, 'conv', (1, 1), plain=True) for m in (self.linear, self.fc)]

    def forward_hip(self, x, keep_row, key_keep, hip):
        """x [B, T, C] in the compute dtype; key_keep [B, 1, 1, T] additive bias (0 = attend, -inf = padding); returns the masked sub-layer
        output layer_norm(dropout(fc(attention)) + x) * non_pad_mask"""
        bank, (l_qkv, l_fc) = hip
        bs, T, _ = x.shape
        H, dk, dv = self.n_head, self.d_k, self.d_v
        qkv, x_res = hip_conv(bank, l_qkv, x.unsqueeze(1), tap=True)        # (tap: see PositionwiseFeedForward.forward_hip)
        x = x_res.squeeze(1)
        qkv = qkv.view(bs, T, H, 2 * dk + dv).transpose(1, 2)               # [bs, H, T, 2dk+dv] (a view)
        att = self.attention
        p = att.dropout.p if (hasattr(att, 'dropout') and att.training) else 0.0
        if torch.is_tensor(key_keep):
            q, k, v = _SplitHeads.apply(qkv, dk)
            out = F.scaled_dot_product_attention(q, k, v, attn_mask=key_keep, dropout_p=p, scale=1.0 / att.temperature)
            out = out.transpose(1, 2).reshape(bs, 1, T, H * dv)
        else:       # padded key bias (hip/attn.py): the attention core on the kernels, q / k / v read in place
            out = hipattn.attention(qkv.transpose(1, 2).reshape(bs, T, H * (2 * dk + dv)), key_keep[0], H,
                                    1.0 / att.temperature, p, self._salt_attn).unsqueeze(1)
        pd = self.dropout.p if self.training else 0.0
        # (the output projection runs inside the add + LayerNorm launch where the fused kernel takes the shape)
        return hip_conv_add_ln(bank, l_fc, out, x, self.layer_norm.weight, self.layer_norm.bias, keep_row=keep_row, p_drop=pd,
                               salt=self._salt, eps=self.layer_norm.eps)


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d_in, d_hid, fft_conv1d_kernel, fft_conv1d_padding, dropout, name, fused_layernorm=False):
        super().__init__()
        self.name = name
        self.w_1 = nn.Conv1d(d_in, d_hid, kernel_size=fft_conv1d_kernel, padding=fft_conv1d_padding)
        self.w_2 = nn.Conv1d(d_hid, d_in, kernel_size=fft_conv1d_kernel, padding=fft_conv1d_padding)
        self.layer_norm = nn.LayerNorm(d_in)
        self.dropout = nn.Dropout(dropout)
        self._salt = hipnorm.new_salt()

    def hip_layers(self):
        k, p = self.w_1.kernel_size[0], self.w_1.padding[0]
        return [ConvLayer(m, 'conv', (1, k), (1, 1), (1, 1), (0, p), plain=True) for m in (self.w_1, self.w_2)]

    def forward_hip(self, x, keep_row, hip):
        """x [B, T, C] (already masked) in the compute dtype -> layer_norm(dropout(w_2(relu(w_1 x))) + x) * non_pad_mask"""
        bank, (l1, l2) = hip
        # [B, T, C] IS the channels-last layout of a 1-D convolution: no transposes; the ReLU between the two
        # convolutions is applied once, in the first one's epilogue (leaky slope 0), and its derivative in the second one's
        # data-gradient epilogue
        # (tap: the residual input of the fused LayerNorm reads the alias the first convolution hands back, so the residual
        # gradient is added in that convolution's data-gradient epilogue)
        h1, x_res = hip_conv(bank, l1, x.unsqueeze(1), tap=True, out_slope=0.0, out_masked=True)
        h = hip_conv(bank, l2, h1, in_act=0.0).squeeze(1)
        x = x_res.squeeze(1)
        pd = self.dropout.p if self.training else 0.0
        return hipnorm.add_layer_norm(h, x, self.layer_norm.weight, self.layer_norm.bias, keep_row=keep_row, p_drop=pd,
                                      salt=self._salt, eps=self.layer_norm.eps)


class FFTBlock(nn.Module):
    def __init__(self, d_model, d_inner, n_head, d_k, d_v, fft_conv1d_kernel, fft_conv1d_padding, dropout, name,
                 attn_dropout=0.1, fused_layernorm=False):
        super().__init__()
        self.slf_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout, name + '.slf_attn', attn_dropout)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_inner, fft_conv1d_kernel, fft_conv1d_padding, dropout,
                                               name + '.pos_ffn')

    def forward_hip(self, x, keep_row, key_keep, hip_attn, hip_ffn):
        return self.pos_ffn.forward_hip(self.slf_attn.forward_hip(x, keep_row, key_keep, hip_attn), keep_row, hip_ffn)


class FFTBlocks(nn.Module):
    def __init__(self, max_seq_len, n_layers, n_head, d_k, d_v, d_model, d_inner, fft_conv1d_kernel,
                 fft_conv1d_padding, dropout, name, attn_dropout=0.1, fused_layernorm=False):
        super().__init__()
        self.name, self.d_model, self.max_seq_len = name, d_model, max_seq_len
        self.position = nn.Embedding.from_pretrained(
            get_sinusoid_encoding_table(max_seq_len + 1, d_model, padding_idx=0), freeze=True)
        self.layer_stack = nn.ModuleList([
            FFTBlock(d_model, d_inner, n_head, d_k, d_v, fft_conv1d_kernel, fft_conv1d_padding, dropout,
                     '%s.layer_stack.%d' % (name, i), attn_dropout, fused_layernorm) for i in range(n_layers)])

        self.hip_dtype = torch.float32        # compute dtype of the HIP convolutions (trainer: bfloat16 in bf16 runs)
        self._bank = None

    def _hip(self):
        if self._bank is None:
            self._layers = [(blk.slf_attn.hip_layers(), blk.pos_ffn.hip_layers()) for blk in self.layer_stack]
            self._bank = ConvBank([l for attn, ffn in self._layers for l in attn + ffn])
        return self._bank, self._layers

    def forward(self, seq, pos, return_attns=False, acts=None, lengths=None):
        """``lengths`` (optional, per-utterance frame counts): with MSMC_FFT_PROLOGUE=1 the positions 1 .. len / 0, the
        positional-embedding add, the cast, the row mask and the key-padding bias are ONE launch (hip/norm.py
        fft_prologue) and ``pos`` may be None."""
        if not (seq.is_cuda or _interpreter_bound()):
            raise RuntimeError('FFTBlocks runs on the gfx950 kernels only: move the module to the GPU (tests: bind the interpreter build)')
        if FFT_PROLOGUE and lengths is not None and seq.dtype in (torch.float32, torch.bfloat16):
            att0 = self.layer_stack[0].slf_attn
            if hipattn.supported(self.hip_dtype, att0.d_k, att0.d_v):
                bank, layers = self._hip()
                bank.prepare(self.hip_dtype)
                B, T = seq.shape[0], seq.shape[1]
                out, keep_row, bias = hipnorm.fft_prologue(seq, lengths.to(seq.device), self.position.weight, self.hip_dtype,
                                                           (T + 31) // 32 * 32)
                for layer, (attn, ffn) in zip(self.layer_stack, layers):
                    out = layer.forward_hip(out, keep_row, (bias,), (bank, attn), (bank, ffn))
                return out, keep_row.view(torch.bool).view(B, T, 1)
        if pos is None:
            steps = torch.arange(1, seq.shape[1] + 1, device=seq.device).unsqueeze(0)
            pos = steps * (steps <= lengths.to(seq.device).unsqueeze(1))
        keep = get_non_pad_mask(pos)
        out = seq + self.position(pos)
        bank, layers = self._hip()
        bank.prepare(self.hip_dtype)          # one launch: kernel-layout weights of the 4 x n_layers GEMMs / convolutions
        keep_row = pos.ne(0).to(torch.uint8).reshape(-1)
        # additive key-padding bias, built ONCE per stack in the compute dtype and broadcast over heads and queries (a
        # boolean mask is converted to this by every attention call: a where + fills per layer)
        att0 = self.layer_stack[0].slf_attn
        if hipattn.supported(self.hip_dtype, att0.d_k, att0.d_v):
            key_keep = (hipattn.pad_key_bias(pos),)      # csrc/attn.hip (bf16, head size 64)
        else:                                            # the stock fused operator (fp32 parity runs, other head sizes)
            key_keep = torch.zeros(pos.shape[0], 1, 1, pos.shape[1], dtype=self.hip_dtype, device=pos.device).masked_fill_(
                pos.eq(0).view(pos.shape[0], 1, 1, pos.shape[1]), float('-inf'))
        out = out.to(self.hip_dtype)
        for layer, (attn, ffn) in zip(self.layer_stack, layers):
            out = layer.forward_hip(out, keep_row, key_keep, (bank, attn), (bank, ffn))
        return out, keep


def _interpreter_bound():
    from ...hip import lib
    return lib._host_pointers_ok


class DurationPredictor(nn.Module):
    """conv-ReLU-LN-dropout x2 + linear on the phoneme axis (reference transformer.py:481-534).  The two convolutions run
    on the gfx950 kernels with the ReLU in their epilogue (one small ConvBank); LayerNorm, dropout and the 256 -> 1 linear
    are a few thousand phonemes of work on stock operators."""

    def __init__(self, input_size, filter_size, kernel, dropout, fused_layernorm=False):
        super().__init__()
        self.input_size, self.filter_size, self.kernel, self.dropout = input_size, filter_size, kernel, dropout
        self.conv1d_1 = nn.Conv1d(input_size, filter_size, kernel_size=kernel, padding=1)
        self.relu_1 = nn.ReLU()
        self.layer_norm_1 = nn.LayerNorm(filter_size)
        self.dropout_1 = nn.Dropout(dropout)
        self.conv1d_2 = nn.Conv1d(filter_size, filter_size, kernel_size=kernel, padding=1)
        self.relu_2 = nn.ReLU()
        self.layer_norm_2 = nn.LayerNorm(filter_size)
        self.dropout_2 = nn.Dropout(dropout)
        self.linear_layer = nn.Linear(filter_size, 1, bias=True)
        self.hip_dtype = torch.float32        # compute dtype of the two convolutions (trainer: bfloat16 in bf16 runs)
        self._bank = None

    def _conv_relu(self, conv, x, hip):
        if not hip:
            return F.relu(conv(x.transpose(1, 2)).transpose(1, 2))
        return hip_conv(self._bank, self._layers[id(conv)], x.to(self.hip_dtype).contiguous().unsqueeze(1), out_slope=0.0).squeeze(1)

    def forward(self, input, input_mask):
        out = input * input_mask.to(input.dtype)
        hip = out.is_cuda or _interpreter_bound()
        if hip:
            if self._bank is None:
                self._layers = {id(m): ConvLayer(m, 'conv', (1, m.kernel_size[0]), (1, 1), (1, 1), (0, m.padding[0]), plain=True)
                                for m in (self.conv1d_1, self.conv1d_2)}
                self._bank = ConvBank(list(self._layers.values()))
            self._bank.prepare(self.hip_dtype)
        out = self.dropout_1(self.layer_norm_1(self._conv_relu(self.conv1d_1, out, hip)))
        out = self.dropout_2(self.layer_norm_2(self._conv_relu(self.conv1d_2, out, hip)))
        out = self.linear_layer(out) * input_mask.to(out.dtype)
        return out.squeeze(-1)


class LengthRegulator(nn.Module):
    """Phoneme -> frame expansion by (target or predicted) durations (reference transformer.py:427-478)."""

    def __init__(self, input_size, duration_predictor_filter_size, duration_predictor_kernel_size, dropout,
                 fused_layernorm=False):
        super().__init__()
        self.duration_predictor = DurationPredictor(input_size, duration_predictor_filter_size,
                                                    duration_predictor_kernel_size, dropout, fused_layernorm)

    def forward(self, input, input_mask, target=None, alpha=1.0, width=None):
        duration = self.duration_predictor(input, input_mask)
        if self.training:
            output, pos = self.get_output(input, target, alpha, width)
            return output, pos, duration
        duration = torch.clamp_min(duration, 0) if target is None else target
        output, pos = self.get_output(input, duration, alpha, width)
        return output, pos, torch.round(duration).long()

    def get_output(self, input, duration, alpha, width=None):
        """one gather for the whole batch: frame t of utterance b copies phoneme searchsorted(cumsum(repeats_b), t);
        zero-padded to the longest utterance, positions 1..len (0 on padding) like the reference's pad_sequence.
        ``width``: the caller's frame count (a host integer at least as large as the longest utterance): no device read-back,
        the step stays hipGraph-capturable; the extra frames are padding (zeros, position 0) like the rest."""
        repeats = torch.round(duration.float() * alpha).long().clamp_min(0)
        ends = repeats.cumsum(1)                                              # [B, P]
        total = ends[:, -1]
        if width is None:
            width = int(total.max())
        t = torch.arange(width, device=input.device).unsqueeze(0).expand(input.shape[0], -1)
        src = torch.searchsorted(ends, t.contiguous(), right=True).clamp_max(input.shape[1] - 1)
        valid = t < total.unsqueeze(1)
        output = torch.gather(input, 1, src.unsqueeze(-1).expand(-1, -1, input.shape[2])) * valid.unsqueeze(-1).to(input.dtype)
        pos = (t + 1) * valid.long()
        return output, pos
