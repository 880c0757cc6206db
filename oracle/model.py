"""Oracle: the MSMC-VQ-GAN autoencoder and UnivNet discriminator as pure functions over a
parameter dictionary keyed by the reference's ``state_dict`` names.  TEST INFRASTRUCTURE ONLY.

Plain PyTorch fp32 on CPU.  Restates
  * FFTBlocks / FFTBlock / MultiHeadAttention / PositionwiseFeedForward
        reference msmctts/networks/acoustic_models/transformer.py:71-424
  * MultiStageEncoder, PriorPredictor, MultiStageQuantizer, MSMCVQGAN
        reference msmctts/networks/vqgantts/msmc_vqgan.py:14-350
  * ResStack + gate                       reference msmctts/networks/vqgantts/modules.py:172-251
  * Generator, ResBlock1                  reference msmctts/networks/hifigan/generator.py:10-55, common.py:21-51
  * DiscriminatorR/P, MRD, MPD, wrapper   reference msmctts/networks/hifigan/discriminator.py:15-190
``P`` maps names such as ``autoencoder.decoder.ups.0.weight_v`` to tensors; VQ buffers in ``P``
are updated in place exactly where the reference updates them.
"""
import math

import torch
import torch.nn.functional as F

from . import audio
from .vq import multi_head_quantize


# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------
def pad_mask(lengths, max_len=None):
    """True where t >= length (utils/utils.py:154-158)."""
    max_len = int(lengths.max()) if max_len is None else max_len
    return ~(torch.arange(max_len, device=lengths.device)[None, :] < lengths[:, None])


def wn_weight(P, name):
    """Old-style weight norm: w = g * v / ||v||, norm over all dims but 0."""
    v, g = P[name + '.weight_v'], P[name + '.weight_g']
    return torch._weight_norm(v, g, 0)


def _drop(x, p, training):
    return F.dropout(x, p, training) if (training and p > 0) else x


def position_ids(lengths, width):
    pos = torch.arange(1, width + 1, device=lengths.device)[None, :].repeat(lengths.shape[0], 1)
    return pos.masked_fill(pad_mask(lengths, width), 0).long()


# ----------------------------------------------------------------------------
# FFT blocks (transformer.py)
# ----------------------------------------------------------------------------
def fft_blocks(P, pre, x, pos, cfg, training):
    """x (B, T, d_model), pos (B, T) with 0 = padding.  transformer.py:123-146."""
    n_head, d_k, d_v = cfg['n_head'], cfg['d_k'], cfg['d_v']
    p_drop, p_attn = cfg.get('dropout', 0.0), cfg.get('attn_dropout', 0.1)
    pad_k = cfg['fft_conv1d_padding']
    key_pad = pos.eq(0)                       # (B, T)
    keep = pos.ne(0).unsqueeze(-1).to(x.dtype)
    out = x + F.embedding(pos, P[pre + '.position.weight'])
    B, T, _ = out.shape
    for l in range(cfg['n_layers']):
        lp = '%s.layer_stack.%d' % (pre, l)
        # --- self attention, post-LN (transformer.py:237-288)
        res = out
        qkv = F.linear(out, P[lp + '.slf_attn.linear.weight'], P[lp + '.slf_attn.linear.bias'])
        qkv = qkv.view(B, T, n_head, 2 * d_k + d_v).permute(2, 0, 1, 3).reshape(n_head * B, T, -1)
        q, k, v = qkv[..., :d_k], qkv[..., d_k:2 * d_k], qkv[..., 2 * d_k:]
        att = torch.bmm(q, k.transpose(1, 2)) / math.sqrt(d_k)
        att = att.masked_fill(key_pad.unsqueeze(1).expand(-1, T, -1).repeat(n_head, 1, 1), -math.inf)
        att = _drop(torch.softmax(att, dim=2), p_attn, training)
        ctx = torch.bmm(att, v).view(n_head, B, T, d_v).permute(1, 2, 0, 3).reshape(B, T, n_head * d_v)
        ctx = F.linear(ctx, P[lp + '.slf_attn.fc.weight'], P[lp + '.slf_attn.fc.bias'])
        out = _drop(ctx, p_drop, training) + res
        out = F.layer_norm(out, (out.shape[-1],), P[lp + '.slf_attn.layer_norm.weight'],
                           P[lp + '.slf_attn.layer_norm.bias'])
        out = out * keep
        # --- conv feed-forward (transformer.py:357-385)
        res = out
        h = F.conv1d(out.transpose(1, 2), P[lp + '.pos_ffn.w_1.weight'], P[lp + '.pos_ffn.w_1.bias'],
                     padding=pad_k)
        h = F.conv1d(F.relu(h), P[lp + '.pos_ffn.w_2.weight'], P[lp + '.pos_ffn.w_2.bias'],
                     padding=pad_k).transpose(1, 2)
        out = _drop(h, p_drop, training) + res
        out = F.layer_norm(out, (out.shape[-1],), P[lp + '.pos_ffn.layer_norm.weight'],
                           P[lp + '.pos_ffn.layer_norm.bias'])
        out = out * keep
    return out


# ----------------------------------------------------------------------------
# encoder / quantiser (msmc_vqgan.py)
# ----------------------------------------------------------------------------
def multi_stage_encoder(P, pre, x, lengths, cfg, training):
    """-> [(feat, length)] fine -> coarse, stages chained (msmc_vqgan.py:46-62)."""
    outs = []
    feat, flen = x, lengths
    for i, scale in enumerate(cfg['downsample_scales']):
        if scale > 1:
            feat = F.avg_pool1d(feat.transpose(1, 2), kernel_size=scale, stride=scale,
                                ceil_mode=True).transpose(1, 2)
            flen = torch.ceil(flen / scale).int()
        pos = position_ids(flen, int(flen.max()))
        feat = fft_blocks(P, '%s.encoders.%d' % (pre, i), feat, pos, cfg, training)
        outs.append((feat, flen))
    return outs


# ResStack hard-codes p_dropout=0.1 (modules.py:183); the golden fixtures zero every nn.Dropout, so the
# parity tests set this module attribute to 0.0.
RESSTACK_DROPOUT = 0.1


def res_stack(P, pre, x, x_mask, n_layers, kernel_size, dilation_rate, training, p_dropout=None):
    """WaveNet stack without conditioning (modules.py:229-251)."""
    p_dropout = RESSTACK_DROPOUT if p_dropout is None else p_dropout
    C = x.shape[1]
    output = torch.zeros_like(x)
    for i in range(n_layers):
        dil = dilation_rate ** i
        padding = int((kernel_size * dil - dil) / 2)
        x_in = F.conv1d(x, wn_weight(P, '%s.in_layers.%d' % (pre, i)),
                        P['%s.in_layers.%d.bias' % (pre, i)], dilation=dil, padding=padding)
        acts = torch.tanh(x_in[:, :C]) * torch.sigmoid(x_in[:, C:])
        acts = _drop(acts, p_dropout, training)
        rs = F.conv1d(acts, wn_weight(P, '%s.res_skip_layers.%d' % (pre, i)),
                      P['%s.res_skip_layers.%d.bias' % (pre, i)])
        if i < n_layers - 1:
            x = (x + rs[:, :C]) * x_mask
            output = output + rs[:, C:]
        else:
            output = output + rs
    return output * x_mask


def prior_predictor(P, pre, x, lengths, prior_cfg, training):
    """-> (hidden, projected) both (B, T, C)  (msmc_vqgan.py:83-88)."""
    xt = x.transpose(1, 2)
    x_mask = (~pad_mask(lengths, xt.shape[2])).unsqueeze(1).to(x.dtype)
    h = res_stack(P, pre + '.enc', xt, x_mask, prior_cfg.get('n_layers', 4),
                  prior_cfg.get('kernel_size', 5), prior_cfg.get('dilation_rate', 1), training)
    o = F.conv1d(h, P[pre + '.proj.weight'], P[pre + '.proj.bias']) * x_mask
    return h.transpose(1, 2), o.transpose(1, 2)


def _vq_heads(P, pre, n_heads):
    if n_heads == 1:
        return [(P[pre + '.embed'], P[pre + '.cluster_size'], P[pre + '.embed_avg'])]
    return [(P['%s.quantizers.%d.embed' % (pre, h)], P['%s.quantizers.%d.cluster_size' % (pre, h)],
             P['%s.quantizers.%d.embed_avg' % (pre, h)]) for h in range(n_heads)]


def multi_stage_quantizer(P, pre, encoder_states, scales_down, qcfg, training):
    """msmc_vqgan.py:147-234 with upsampling='repeat' (all shipped configs)."""
    assert qcfg.get('upsampling', 'repeat') == 'repeat'
    n_heads = qcfg.get('n_heads', 4)
    p_drop = qcfg.get('dropout', 0.1)
    update = qcfg.get('update_codebook', True)
    up = list(scales_down)[::-1]
    states = encoder_states[::-1]
    residual = None
    quants, diffs, inds, preds = [], [], [], []
    for i, (emb, length) in enumerate(states):
        if residual is None:
            pred_q = None
        else:
            residual = residual[:, :int(length.max())]
            hid, pred_q = prior_predictor(P, '%s.predictor.%d' % (pre, i), residual, length,
                                          qcfg.get('prior_config', {}), training)
            residual = residual + _drop(hid, p_drop, training)
        pre_in = emb if residual is None else torch.cat((emb, residual), dim=-1)
        h = pre_in.transpose(1, 2)
        h = F.conv1d(h, P['%s.preprocessor.%d.0.weight' % (pre, i)], P['%s.preprocessor.%d.0.bias' % (pre, i)])
        h = F.conv1d(torch.tanh(h), P['%s.preprocessor.%d.2.weight' % (pre, i)],
                     P['%s.preprocessor.%d.2.bias' % (pre, i)])
        assert not qcfg.get('norm', False), 'BatchNorm variant not used by shipped configs'
        q_in = h.transpose(1, 2)
        q, dff, ind = multi_head_quantize(q_in, length, _vq_heads(P, '%s.quantizer.%d' % (pre, i), n_heads),
                                          training and update)
        if n_heads == 1:
            ind = ind.squeeze(-1)
        post_in = q if residual is None else torch.cat((residual, q), dim=-1)
        h = F.linear(post_in, P['%s.postprocessor.%d.0.weight' % (pre, i)],
                     P['%s.postprocessor.%d.0.bias' % (pre, i)])
        h = F.linear(torch.tanh(h), P['%s.postprocessor.%d.2.weight' % (pre, i)],
                     P['%s.postprocessor.%d.2.bias' % (pre, i)])
        h = _drop(h, p_drop, training)
        residual = h if residual is None else residual + h
        quants.append(q)
        diffs.append(dff)
        inds.append(ind)
        preds.append((pred_q, q, length))
        residual = torch.repeat_interleave(residual, up[i], dim=1)
    out = {'residual_output': residual, 'quantizer_outputs': quants, 'quantizer_diffs': diffs,
           'quantizer_indices': inds, 'quantizer_lengths': [s[1] for s in states],
           'predictor_diffs': None}
    if training:
        loss = {'total_loss': 0}
        for i, (p, t, length) in enumerate(preds):
            if p is None:
                continue
            l = F.mse_loss(p, t.detach(), reduction='none').mean(-1)
            l = l.masked_fill(pad_mask(length, l.shape[1]), 0).sum() / sum(length)
            loss['embed_loss_mse_%d' % i] = l
            loss['total_loss'] = loss['total_loss'] + l * 1.0
        out['predictor_diffs'] = loss
    return out


# ----------------------------------------------------------------------------
# HifiGAN generator (generator.py:40-55, common.py:44-51)
# ----------------------------------------------------------------------------
def hifigan_generator(P, pre, x, dcfg):
    """x (B, C_in, T) -> (B, 1, T * prod(upsample_rates))."""
    ks, dils = dcfg['resblock_kernel_sizes'], dcfg['resblock_dilation_sizes']
    nk = len(ks)
    x = F.conv1d(x, wn_weight(P, pre + '.conv_pre'), P[pre + '.conv_pre.bias'], padding=3)
    for i, (u, k) in enumerate(zip(dcfg['upsample_rates'], dcfg['upsample_kernel_sizes'])):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, wn_weight(P, '%s.ups.%d' % (pre, i)), P['%s.ups.%d.bias' % (pre, i)],
                               stride=u, padding=(k - u) // 2)
        acc = None
        for j in range(nk):
            rp = '%s.resblocks.%d' % (pre, i * nk + j)
            y = x
            for m, d in enumerate(dils[j]):
                t = F.leaky_relu(y, 0.1)
                t = F.conv1d(t, wn_weight(P, '%s.convs1.%d' % (rp, m)), P['%s.convs1.%d.bias' % (rp, m)],
                             dilation=d, padding=int((ks[j] * d - d) / 2))
                t = F.leaky_relu(t, 0.1)
                t = F.conv1d(t, wn_weight(P, '%s.convs2.%d' % (rp, m)), P['%s.convs2.%d.bias' % (rp, m)],
                             padding=int((ks[j] - 1) / 2))
                y = t + y
            acc = y if acc is None else acc + y
        x = acc / nk
    x = F.leaky_relu(x)                          # slope 0.01 (generator.py:52)
    x = F.conv1d(x, wn_weight(P, pre + '.conv_post'), P[pre + '.conv_post.bias'], padding=3)
    return torch.tanh(x)


# ----------------------------------------------------------------------------
# autoencoder (msmc_vqgan.py:309-350)
# ----------------------------------------------------------------------------
def msmc_vqgan_forward(P, cfg, mel, mel_length, warmup=False, window=None, training=True,
                       pre='autoencoder'):
    ecfg, qcfg = cfg['encoder_config'], cfg['quantizer_config']
    x = F.linear(mel, P[pre + '.in_linear.weight'], P[pre + '.in_linear.bias'])
    enc = multi_stage_encoder(P, pre + '.encoder', x, mel_length, ecfg, training)
    qs = multi_stage_quantizer(P, pre + '.quantizer', enc, ecfg['downsample_scales'], qcfg, training)
    dec_in = qs['residual_output']
    out = {'encoder_outputs': [e[0] for e in enc][::-1], 'encoder_lengths': [e[1] for e in enc][::-1],
           'encoder_indices': qs['quantizer_indices'], 'encoder_diffs': qs['quantizer_diffs'],
           'decoder_diffs': qs['predictor_diffs']}
    if cfg.get('frame_decoder_config') is not None:
        pos = position_ids(mel_length, int(mel_length.max()))
        dec_in = fft_blocks(P, pre + '.frame_decoder', dec_in, pos, cfg['frame_decoder_config'], training)
    if cfg.get('pred_mel', False):
        out['mel_outputs'] = F.linear(dec_in, P[pre + '.mel_predictor.weight'], P[pre + '.mel_predictor.bias'])
    if not warmup:
        if window is not None:
            assert len(window) == dec_in.shape[0]
            dec_in = torch.stack([dec_in[i, s:e] for i, (s, e) in enumerate(window)], dim=0)
        wav = hifigan_generator(P, pre + '.decoder', dec_in.transpose(1, 2), cfg['decoder_config'])
        out['decoder_outputs'] = wav.transpose(1, 2)
    return out


# ----------------------------------------------------------------------------
# discriminator (discriminator.py)
# ----------------------------------------------------------------------------
_MRD_STRIDES = (1, 2, 1, 2, 1, 2, 1)


def discriminator_r(P, pre, img):
    """img (B, 2, F, T') -> (score, 6 fmaps).  The reference's in-place LeakyReLU aliases the stored
    fmaps (discriminator.py:28,71-76): fmap_j == leaky_relu(conv_j(.), 0.2) for j < 6."""
    fmaps = []
    x = img
    for j, s in enumerate(_MRD_STRIDES):
        name = '%s.discriminator.%d.%d' % (pre, j, 1 if j == 0 else 2)
        if j > 0:
            x = F.leaky_relu(x, 0.2)
            fmaps.append(x)
        x = F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), wn_weight(P, name), P[name + '.bias'], stride=s)
    return x, fmaps


def discriminator_p(P, pre, y, period):
    """y (B, 1, L) -> (flat score, 5 pre-activation fmaps)  (discriminator.py:135-154)."""
    b, c, t = y.shape
    if t % period != 0:
        n_pad = period - (t % period)
        y = F.pad(y, (0, n_pad), 'reflect')
        t = t + n_pad
    x = y.view(b, c, t // period, period)
    fmaps = []
    for j in range(5):
        name = '%s.convs.%d' % (pre, j)
        stride = (3, 1) if j < 4 else (1, 1)
        x = F.conv2d(x, wn_weight(P, name), P[name + '.bias'], stride=stride, padding=(2, 0))
        fmaps.append(x)
        x = F.leaky_relu(x, 0.2)
    x = F.conv2d(x, wn_weight(P, pre + '.conv_post'), P[pre + '.conv_post.bias'], padding=(1, 0))
    return torch.flatten(x, 1, -1), fmaps


def discriminator_forward(P, dcfg, y, pre='discriminator'):
    """y (B, L) or (B, 1, L) -> (scores, fmaps): MRD entries first, then MPD (discriminator.py:180-190)."""
    if y.dim() == 2:
        y = y.unsqueeze(1)
    mrd, mpd = dcfg['mrd_config'], dcfg['mpd_config']
    assert mrd.get('domain', 'double') == 'double'
    scores, fmaps = [], []
    for i, hop in enumerate(mrd['hop_lengths']):
        img = audio.mrd_spectrogram(y.squeeze(1), hop, mrd.get('sample_rate', 24000),
                                    mrd.get('mel_scale', True))
        s, f = discriminator_r(P, '%s.mrd.discriminators.%d' % (pre, i), img)
        scores.append(s)
        fmaps.append(f)
    for i, p in enumerate(mpd['periods']):
        s, f = discriminator_p(P, '%s.mpd.discriminators.%d' % (pre, i), y, p)
        scores.append(s)
        fmaps.append(f)
    return scores, fmaps
