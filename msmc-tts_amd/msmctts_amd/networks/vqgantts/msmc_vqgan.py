"""MSMC-VQ-GAN autoencoder (drop-in for reference msmctts/networks/vqgantts/msmc_vqgan.py:14-410).

Same classes, constructor kwargs (the YAML surface), ``state_dict`` keys and output dictionary.
The quantiser calls the gfx950 VQ kernels; the vocoder is ``HifiGANGenerator``.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...hip import losses as hiploss
from ...hip import norm as hipnorm
from ...hip import vq as hipvq
from ...hip.convnet import ConvBank, ConvLayer, hip_conv
from ...utils.utils import get_mask_from_lengths
from ..acoustic_models.transformer import FFTBlocks
from ..hifigan.generator import Generator as HifiGANGenerator
from .modules import MultiHeadQuantize, Quantize, ResStack


def _positions(lengths, device, width):
    """1..len per utterance, 0 on padding (msmc_vqgan.py:56-58).  The reference takes the width from
    ``lengths.max()`` (a host sync); the batch contract guarantees it equals the padded length."""
    pos = torch.arange(1, width + 1, device=device).unsqueeze(0).repeat(lengths.shape[0], 1)
    return pos.masked_fill(get_mask_from_lengths(lengths.to(device), width), 0)


def _fft_pos(lengths, feat):
    """the positions an FFT stack is called with; None when the stack derives them from ``lengths`` itself
    (MSMC_FFT_PROLOGUE=1: acoustic_models/transformer.py FFTBlocks.forward)"""
    from ..acoustic_models import transformer
    return None if transformer.FFT_PROLOGUE else _positions(lengths, feat.device, feat.shape[1])


class MultiStageEncoder(nn.Module):
    def __init__(self, in_channels, downsample_scales=[1], max_seq_len=2400, n_layers=4, n_head=2, d_k=64, d_v=64,
                 d_inner=1024, fft_conv1d_kernel=3, fft_conv1d_padding=1, dropout=0.2, attn_dropout=0.1,
                 fused_layernorm=False):
        super().__init__()
        self.downsample_scales = list(downsample_scales)
        self.encoders = nn.ModuleList([
            FFTBlocks(max_seq_len=max_seq_len, n_layers=n_layers, n_head=n_head, d_k=d_k, d_v=d_v,
                      d_model=in_channels, d_inner=d_inner, fft_conv1d_kernel=fft_conv1d_kernel,
                      fft_conv1d_padding=fft_conv1d_padding, dropout=dropout, attn_dropout=attn_dropout,
                      fused_layernorm=fused_layernorm, name='encoder_%d' % i)
            for i in range(len(self.downsample_scales))])

    def forward(self, input, input_length):
        outputs = []
        feat, flen = input, input_length
        for enc, scale in zip(self.encoders, self.downsample_scales):
            if scale > 1:                          # stages are chained: pool the previous stage's output
                feat = F.avg_pool1d(feat.transpose(1, 2), kernel_size=scale, stride=scale,
                                    ceil_mode=True).transpose(1, 2)
                flen = torch.ceil(flen / scale).int()
            feat, _ = enc(feat, _fft_pos(flen, feat), lengths=flen)
            outputs.append((feat, flen))
        return outputs


def _pointwise(module, x):
    """``module`` (a kernel-size-1 ``nn.Conv1d``, kept for its checkpoint keys) applied to channels-last x (B, T, C):
    a 1x1 convolution IS a linear layer on the last axis -- no transposes, no NCHW convolution kernel."""
    return F.linear(x, module.weight.squeeze(-1), module.bias)


def _pointwise_stack(seq, x):
    """nn.Sequential of 1x1 Conv1d / Tanh (reference pre-processor, msmc_vqgan.py:115-136) on channels-last x;
    falls back to the channels-first modules for anything else (the optional BatchNorm1d)."""
    mods = list(seq)
    if all(isinstance(m, nn.Tanh) or (isinstance(m, nn.Conv1d) and m.kernel_size == (1,)) for m in mods):
        for m in mods:
            x = torch.tanh(x) if isinstance(m, nn.Tanh) else _pointwise(m, x)
        return x
    return seq(x.transpose(1, 2)).transpose(1, 2)


class PriorPredictor(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=5, dilation_rate=1, n_layers=4):
        super().__init__()
        self.enc = ResStack(in_channels, kernel_size, dilation_rate, n_layers)
        self.proj = nn.Conv1d(in_channels, out_channels, 1)
        self._salts = [hipnorm.new_salt() for _ in range(n_layers)]

    def forward(self, x, x_lengths):
        x = x.transpose(1, 2)
        x_mask = (~get_mask_from_lengths(x_lengths.to(x.device), x.shape[2])).unsqueeze(1).to(x.dtype)
        h = self.enc(x, x_mask).transpose(1, 2)
        return h, _pointwise(self.proj, h) * x_mask.transpose(1, 2)

    def hip_layers(self):
        """[(in_layer_i, res_skip_i) ...] + [proj] as layers of the quantiser's ConvBank (channels-last [B,1,T,C])"""
        pairs = [(a.hip_layer(), b.hip_layer()) for a, b in zip(self.enc.in_layers, self.enc.res_skip_layers)]
        return pairs, ConvLayer(self.proj, 'conv', (1, 1), plain=True)

    def forward_hip(self, x, x_lengths, hip):
        """x [B, T, C] in the compute dtype -> (hidden, projection): the WaveNet stack of vqgantts/modules.py:229-251 with
        its convolutions on the implicit-GEMM kernels and tanh * sigmoid (+ dropout) as one kernel per layer"""
        bank, (pairs, l_proj) = hip
        C = self.enc.hidden_channels
        # [B, T, 1]: 1 on frames, 0 on padding -- ~get_mask_from_lengths cast to the compute dtype, one launch
        keep = hipnorm.row_mask(x_lengths.to(x.device), x.shape[1], x.dtype).unsqueeze(-1)
        x4 = x.unsqueeze(1)
        out = None
        pd = self.enc.drop.p if self.training else 0.0
        for i, (l_in, l_rs) in enumerate(pairs):
            acts = hipnorm.gate(hip_conv(bank, l_in, x4), p_drop=pd, salt=self._salts[i])
            rs = hip_conv(bank, l_rs, acts)
            if i < len(pairs) - 1:
                x4 = (x4 + rs[..., :C]) * keep.unsqueeze(1)
                skip = rs[..., C:]
            else:
                skip = rs
            out = skip if out is None else out + skip
        h = (out * keep.unsqueeze(1)).contiguous()
        return h.squeeze(1), hip_conv(bank, l_proj, h).squeeze(1) * keep


def _interpreter_bound():
    from ...hip import lib
    return lib._host_pointers_ok


class MultiStageQuantizer(nn.Module):
    def __init__(self, n_model_size, upsample_scales, embedding_sizes=512, embedding_dims=256, n_heads=4,
                 prior_config={}, norm=False, upsampling='repeat', dropout=0.1, update_codebook=True):
        super().__init__()
        self.upsample_scales, self.upsampling = list(upsample_scales), upsampling
        self.dropout, self.update_codebook = dropout, update_codebook
        self.quantizer, self.predictor = nn.ModuleList(), nn.ModuleList()
        self.preprocessor, self.postprocessor = nn.ModuleList(), nn.ModuleList()
        if upsampling != 'repeat':
            self.transposed_conv = nn.ModuleList()
        for i, u in enumerate(self.upsample_scales):
            width = n_model_size * (1 if i == 0 else 2)
            self.predictor.append(PriorPredictor(n_model_size, embedding_dims, **prior_config))
            pre = [nn.Conv1d(width, embedding_dims, 1), nn.Tanh(), nn.Conv1d(embedding_dims, embedding_dims, 1)]
            if norm:
                pre.append(nn.BatchNorm1d(embedding_dims, eps=1e-05, affine=False))
            self.preprocessor.append(nn.Sequential(*pre))
            self.quantizer.append(Quantize(embedding_dims, embedding_sizes) if n_heads == 1 else
                                  MultiHeadQuantize(embedding_dims, embedding_sizes, n_heads))
            self.postprocessor.append(nn.Sequential(
                nn.Linear(embedding_dims * (1 if i == 0 else 2), embedding_dims), nn.Tanh(),
                nn.Linear(embedding_dims, n_model_size)))
            if upsampling != 'repeat':
                k = u * 2 if u % 2 == 0 else u * 2 + 1
                self.transposed_conv.append(nn.ConvTranspose1d(n_model_size, n_model_size, k, u, padding=(k - u) // 2))
        self.hip_dtype = torch.float32        # compute dtype of the HIP GEMMs (trainer: bfloat16 in bf16 runs)
        self.use_hip = not norm            # (the normalised variant -- no shipped configuration uses it -- keeps stock operators)
        self._bank = None

    # -- the 1x1 channel GEMMs, the prior predictor's WaveNet stack and their Tanh / gate on the gfx950 kernels ---------
    def _hip(self):
        if self._bank is None:
            k1 = lambda m: ConvLayer(m, 'conv', (1, 1), plain=True)
            self._stages = []
            for pre, post, pred in zip(self.preprocessor, self.postprocessor, self.predictor):
                self._stages.append(((k1(pre[0]), k1(pre[2])), (k1(post[0]), k1(post[2])), pred.hip_layers()))
            flat = []
            for pre, post, (pairs, proj) in self._stages:
                flat += list(pre) + list(post) + [l for pr in pairs for l in pr] + [proj]
            self._bank = ConvBank(flat)
        return self._bank, self._stages

    def _dropout_add(self, x, res, stage, site):
        """F.dropout(x, self.dropout) + res (res None: the dropout alone) -- reference msmc_vqgan.py:141-176; on the kernels'
        dtypes one launch with the counter-hash masks of hip/norm.py (two stock launches and a stored mask otherwise)"""
        p = self.dropout if self.training else 0.0
        if self.use_hip and p > 0 and hipnorm.dropout_add_usable(x, res):
            if not hasattr(self, '_salts'):
                self._salts = {}
            salt = self._salts.setdefault((stage, site), hipnorm.new_salt())
            return hipnorm.dropout_add(x, res, p, salt)
        x = F.dropout(x, p=self.dropout, training=self.training)
        return x if res is None else res + x

    @staticmethod
    def _stack_hip(bank, pair, x):
        """conv1x1 -> Tanh -> conv1x1 on channels-last x [B, T, C] (pre- / post-processor, msmc_vqgan.py:115-136)"""
        h = hipnorm.tanh(hip_conv(bank, pair[0], x.unsqueeze(1)))
        return hip_conv(bank, pair[1], h).squeeze(1)

    def forward(self, encoder_states, from_encoder=True):
        states = list(encoder_states)
        if from_encoder:
            states = states[::-1]                   # coarse -> fine
        residual = None
        quants, diffs, inds, preds = [], [], [], []
        dev = next(self.parameters()).device
        hip = None
        if self.use_hip and (dev.type == 'cuda' or _interpreter_bound()):
            bank, stages = self._hip()
            bank.prepare(self.hip_dtype)              # one launch: kernel-layout weights of every GEMM of the quantiser
            hip, dt = (bank, stages), self.hip_dtype
        for i, (emb, length) in enumerate(states):
            if residual is None:
                pred_q = None
            else:
                residual = residual[:, :(emb.shape[1] if emb is not None else int(length.max()))]
                if hip is None:
                    hid, pred_q = self.predictor[i](residual, length)
                else:
                    hid, pred_q = self.predictor[i].forward_hip(residual.to(dt).contiguous(), length, (bank, stages[i][2]))
                residual = self._dropout_add(hid, residual, i, 0)
            if emb is None:
                q_in = pred_q
            elif from_encoder:
                pre_in = emb if residual is None else torch.cat((emb, residual), dim=-1)
                q_in = (_pointwise_stack(self.preprocessor[i], pre_in) if hip is None else
                        self._stack_hip(bank, stages[i][0], pre_in.to(dt).contiguous()))
            else:
                q_in = emb
            quant, dff, ind = self.quantizer[i](q_in, length, update=self.update_codebook)
            post_in = quant if residual is None else torch.cat((residual, quant), dim=-1)
            post = (self.postprocessor[i](post_in) if hip is None else
                    self._stack_hip(bank, stages[i][1], post_in.to(dt).contiguous()))
            residual = self._dropout_add(post, residual, i, 1)
            quants.append(quant)
            diffs.append(dff)
            inds.append(ind)
            preds.append({'predictor_outputs': pred_q, 'target_outputs': quant, 'target_indices': ind,
                          'target_lengths': length})
            if self.upsampling == 'mapping':
                residual = self.transposed_conv[i](residual.transpose(1, 2)).transpose(1, 2)
            elif self.upsampling == 'residual':
                up = self.transposed_conv[i](residual.transpose(1, 2)).transpose(1, 2)
                residual = torch.repeat_interleave(residual, self.upsample_scales[i], dim=1) + \
                    F.dropout(up, p=self.dropout, training=self.training)
            else:
                residual = torch.repeat_interleave(residual, self.upsample_scales[i], dim=1)
        out = {'residual_output': residual, 'quantizer_outputs': tuple(quants), 'quantizer_diffs': tuple(diffs),
               'quantizer_indices': tuple(inds), 'quantizer_lengths': [s[1] for s in states],
               'predictor_diffs': None}
        if self.training:
            out['predictor_diffs'] = self.compute_embedding_loss(preds, methods=['mse'], loss_weights=[1.0])
        return out

    def compute_embedding_loss(self, pred_states, methods=['mse'], loss_weights=[1.0]):
        losses = {}
        parts, pweights = [], []                # total_loss = sum of weights * terms: one launch (hiploss.weighted_sum)
        for i, st in enumerate(pred_states):
            p = st['predictor_outputs']
            if p is None:
                continue
            weights = loss_weights[i] if isinstance(loss_weights[0], (list, tuple)) else loss_weights
            for method, weight in zip(methods, weights):
                lengths = st['target_lengths']
                if (method == 'mse' and p.dim() == 3 and hiploss.usable(p) and p.dtype in (torch.float32, torch.bfloat16)
                        and st['target_outputs'].dtype in (torch.float32, torch.bfloat16) and torch.is_tensor(lengths)
                        and lengths.device == p.device and lengths.dtype in (torch.int32, torch.int64)):
                    # mean over channels, masked sum over frames, / sum of lengths: one fused masked mean (two launches)
                    loss = hiploss.masked_mean(p, lengths, b=st['target_outputs'].detach())
                    losses['embed_loss_%s_%d' % (method, i)] = loss
                    parts.append(loss)
                    pweights.append(weight)
                    continue
                if method == 'mse':
                    loss = F.mse_loss(p, st['target_outputs'].detach(), reduction='none').mean(-1)
                elif method == 'softmax':
                    B, T, D = p.shape
                    loss = F.cross_entropy(p.view(-1, D), st['target_indices'].detach().view(-1),
                                           reduction='none').view(B, T)
                elif method in ('triple', 'triple_mean'):
                    loss = self.quantizer[i].compute_triple_loss(p, st['target_indices'])
                elif method == 'triple_sum':
                    loss = self.quantizer[i].compute_triple_loss(p, st['target_indices'], reduction='sum')
                else:
                    raise NotImplementedError('embedding loss %r' % method)
                loss = loss.masked_fill(get_mask_from_lengths(lengths.to(loss.device), loss.shape[1]), 0)
                loss = loss.sum() / lengths.sum()
                losses['embed_loss_%s_%d' % (method, i)] = loss
                parts.append(loss)
                pweights.append(weight)
        losses['total_loss'] = hiploss.weighted_sum(parts, pweights) if parts else 0
        return losses


class MSMCVQGAN(nn.Module):
    def __init__(self, in_dim, n_model_size, encoder_config=None, quantizer_config=None, frame_decoder_config=None,
                 decoder_config=None, pred_mel=False):
        super().__init__()
        self.in_linear = nn.Linear(in_dim, n_model_size)
        self.encoder = MultiStageEncoder(n_model_size, **encoder_config)
        self.quantizer = MultiStageQuantizer(n_model_size, list(encoder_config['downsample_scales'])[::-1],
                                             **quantizer_config)
        decoder_config = dict(decoder_config)
        decoder_config['num_mels'] = n_model_size
        self.decoder = HifiGANGenerator(**decoder_config)
        if frame_decoder_config is not None:
            self.frame_decoder = FFTBlocks(d_model=n_model_size, name='frame_decoder', **frame_decoder_config)
        if pred_mel:
            self.mel_predictor = nn.Linear(n_model_size, in_dim)
        self.hip_dtype = torch.float32
        self.use_hip = True
        self._bank = None

    def _hip_ready(self, dev):
        """(build and) refresh the kernel-layout weights of in_linear / mel_predictor: one launch per forward"""
        if not (self.use_hip and (dev.type == 'cuda' or _interpreter_bound())):
            return False
        if self._bank is None:
            mods = [self.in_linear] + ([self.mel_predictor] if hasattr(self, 'mel_predictor') else [])
            self._layers = {id(m): ConvLayer(m, 'conv', (1, 1), plain=True) for m in mods}
            self._bank = ConvBank([self._layers[id(m)] for m in mods])
        self._bank.prepare(self.hip_dtype)
        return True

    def _prepare_banks(self, dev, vocoder):
        """the kernel-layout weight images of every network of this pass in ONE refresh (hip/convnet.py prepare_together):
        an optimizer step moved all of them, and each module refreshing its own bank when the pass reaches it is a dozen
        launches on the critical chain"""
        if not (self.use_hip and (dev.type == 'cuda' or _interpreter_bound())):
            return
        from ...hip import convnet
        if self._bank is None:
            self._hip_ready(dev)
        pairs = [(self._bank, self.hip_dtype)] + [(enc._hip()[0], enc.hip_dtype) for enc in self.encoder.encoders]
        if self.quantizer.use_hip:
            pairs.append((self.quantizer._hip()[0], self.quantizer.hip_dtype))
        if hasattr(self, 'frame_decoder'):
            pairs.append((self.frame_decoder._hip()[0], self.frame_decoder.hip_dtype))
        if vocoder:
            pairs.append((self.decoder._hip()[0], self.decoder.hip_dtype))
        convnet.prepare_together(pairs)

    def _linear(self, module, x, hip):
        """in_linear / mel_predictor as 1-tap implicit GEMMs (bias fused, compute dtype in and out)"""
        if not hip:
            return module(x)
        return hip_conv(self._bank, self._layers[id(module)], x.to(self.hip_dtype).contiguous().unsqueeze(1)).squeeze(1)

    def _decode_frames(self, x, lengths):
        if hasattr(self, 'frame_decoder'):
            x, _ = self.frame_decoder(x, _fft_pos(lengths, x), lengths=lengths)
        return x

    def forward(self, mel, mel_length, warmup=False, window=None):
        if self.training:
            hipnorm.advance_seed(mel.device)        # fresh dropout masks for the fused kernels of this step
        self._prepare_banks(mel.device, vocoder=not warmup)
        hip = self._hip_ready(mel.device)
        enc = self.encoder(self._linear(self.in_linear, mel, hip), mel_length)
        with hipvq.ema_side(mel.device):              # (the codebooks' EMA updates: a side branch, joined before this returns)
            qs = self.quantizer(enc)
        feats, lens = zip(*enc)
        out = {'encoder_outputs': feats[::-1], 'encoder_lengths': lens[::-1],
               'encoder_indices': qs['quantizer_indices'], 'encoder_diffs': qs['quantizer_diffs'],
               'decoder_diffs': qs['predictor_diffs']}
        dec_in = self._decode_frames(qs['residual_output'], mel_length)
        if hasattr(self, 'mel_predictor'):
            out['mel_outputs'] = self._linear(self.mel_predictor, dec_in, hip)
        if not warmup:
            if torch.is_tensor(window):            # (B, n_frames) frame indices on the device: graph-replayable
                dec_in = torch.gather(dec_in, 1, window.unsqueeze(-1).expand(-1, -1, dec_in.shape[-1]))
            elif window is not None:
                assert len(window) == dec_in.shape[0]
                dec_in = torch.stack([dec_in[i, s:e] for i, (s, e) in enumerate(window)], dim=0)
            out['decoder_outputs'] = self.decoder(dec_in.transpose(1, 2)).transpose(1, 2)
        if mel.is_cuda:
            hipvq.join_ema(mel.device)
        return out

    def analysis(self, mel, mel_length):
        enc = self.encoder(self.in_linear(mel), mel_length)
        qs = self.quantizer(enc)
        if self.training:
            feats, lens = zip(*enc)
            return {'encoder_outputs': feats[::-1], 'encoder_lengths': lens[::-1],
                    'encoder_indices': qs['quantizer_indices'], 'encoder_diffs': qs['quantizer_diffs'],
                    'decoder_diffs': qs['predictor_diffs'], 'quantizer_states': qs}
        return qs

    def synthesis(self, quantizer_outputs, quantizer_lengths):
        qs = quantizer_outputs
        if not isinstance(quantizer_outputs, dict):
            qs = self.quantizer(zip(quantizer_outputs, quantizer_lengths), from_encoder=False)
        dec_in = self._decode_frames(qs['residual_output'], quantizer_lengths[-1])
        wav = self.decoder(dec_in.transpose(1, 2)).transpose(1, 2)
        if self.training:
            out = {'decoder_outputs': wav}
            if hasattr(self, 'mel_predictor'):
                out['mel_outputs'] = self.mel_predictor(dec_in)
            return out
        return wav

    def compute_embedding_loss(self, quantizer_outputs, quantizer_lengths, quantizer_states, methods=['mse'],
                               loss_weights=[1.0]):
        states = [{'predictor_outputs': quantizer_outputs[i],
                   'target_outputs': quantizer_states['quantizer_outputs'][i],
                   'target_indices': quantizer_states['quantizer_indices'][i],
                   'target_lengths': quantizer_lengths[i]} for i in range(len(quantizer_outputs))]
        return self.quantizer.compute_embedding_loss(states, methods, loss_weights)
