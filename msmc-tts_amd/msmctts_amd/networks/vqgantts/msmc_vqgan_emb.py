"""MSMC-VQ-GAN over self-supervised speech embeddings -- the QS-TTS synthesiser (drop-in for reference
msmctts/networks/vqgantts/msmc_vqgan_emb.py:14-291; configuration examples/qs-tts/configs/synthesizer/
msmc_vq_gan_hubertch_aishell3.yaml: 1024-dimensional HuBERT frames in, 16 kHz waveform out).

Same classes, constructor kwargs, ``state_dict`` keys and output dictionaries as the reference.  The reference file imports
a module that is not in its tree (``msmc_vqgan_speech``, :11); by its use of ``ResStack`` and ``MultiStageQuantizer`` it is
the MSMC-VQ-GAN module under another name, which is what this file builds on: the multi-stage
quantiser (gfx950 VQ kernels, 1x1 stacks and prior predictor on the implicit-GEMM kernels), the FFT-block stacks and the
HifiGAN generator are the ones of ``msmc_vqgan.py``.  New here: ``MAMSEncoder`` (the multi-stage encoder with an optional
pitch / energy side encoder added to every stage's output) and the optional reference-encoder slot.

The speaker / style reference encoder (``global_encoder_config._name == 'ECAPA_TDNN'``, reference tdnn.py) is not built: the
shipped configuration does not use it; asking for it raises.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...hip import norm as hipnorm
from ..acoustic_models.transformer import FFTBlocks
from ..hifigan.generator import Generator as HifiGANGenerator
from .msmc_vqgan import MultiStageQuantizer, PriorPredictor, _fft_pos


class AttrPredictor(PriorPredictor):
    """reference msmc_vqgan_emb.py:14-38: the prior predictor's WaveNet stack + 1x1 projection under another name"""


class MAMSEncoder(nn.Module):
    """multi-stage FFT-block encoder (reference :41-120): stage i average-pools the previous stage's output by
    ``downsample_scales[i]``; when pitch / energy tracks are given their encoding (a small Conv1d / Tanh stack, pooled
    alongside) is ADDED to every stage's output -- after the first stage's output has been set aside as the content
    representation"""

    def __init__(self, in_channels, pitch_dim=1, energy_dim=1, downsample_scales=[1], max_seq_len=2400, n_layers=4, n_head=2,
                 d_k=64, d_v=64, d_inner=1024, fft_conv1d_kernel=3, fft_conv1d_padding=1, dropout=0.2, attn_dropout=0.1,
                 fused_layernorm=False):
        super().__init__()
        self.downsample_scales = list(downsample_scales)
        self.encoders = nn.ModuleList([
            FFTBlocks(max_seq_len=max_seq_len, n_layers=n_layers, n_head=n_head, d_k=d_k, d_v=d_v, d_model=in_channels,
                      d_inner=d_inner, fft_conv1d_kernel=fft_conv1d_kernel, fft_conv1d_padding=fft_conv1d_padding,
                      dropout=dropout, attn_dropout=attn_dropout, fused_layernorm=fused_layernorm, name='encoder_%d' % i)
            for i in range(len(self.downsample_scales))])
        self.use_pitch = pitch_dim + energy_dim > 0
        if self.use_pitch:
            self.pitch_encoder = nn.Sequential(
                nn.Conv1d(pitch_dim + energy_dim, in_channels, 7, padding=3), nn.Tanh(),
                nn.Conv1d(in_channels, in_channels, 3, padding=1), nn.Tanh(),
                nn.Conv1d(in_channels, in_channels, 3, padding=1), nn.Tanh(),
                nn.Conv1d(in_channels, in_channels, 1))

    def forward(self, emb, input_length, pitch=None, energy=None):
        if self.use_pitch:
            side = self.pitch_encoder(torch.cat((pitch, energy), dim=-1).transpose(1, 2).float()).transpose(1, 2)
        outputs, content = [], None
        feat, flen = emb, input_length
        for enc, scale in zip(self.encoders, self.downsample_scales):
            if scale > 1:
                feat = F.avg_pool1d(feat.transpose(1, 2), kernel_size=scale, stride=scale, ceil_mode=True).transpose(1, 2)
                if self.use_pitch:
                    side = F.avg_pool1d(side.transpose(1, 2), kernel_size=scale, stride=scale, ceil_mode=True).transpose(1, 2)
                flen = torch.ceil(flen / scale).int()
            feat, _ = enc(feat, _fft_pos(flen, feat), lengths=flen)
            if not outputs:
                content = feat
            if self.use_pitch:
                feat = feat + side.to(feat.dtype)
            outputs.append((feat, flen))
        return outputs, content


class MSMCVQGANEmb(nn.Module):
    def __init__(self, emb_dim, n_model_size, pitch_dim=1, energy_dim=1, encoder_config=None, quantizer_config=None,
                 global_encoder_config=None, frame_decoder_config=None, decoder_config=None, pred_mel=False, mel_dim=None):
        super().__init__()
        self.in_linear = nn.Linear(emb_dim, n_model_size)
        self.encoder = MAMSEncoder(n_model_size, pitch_dim=pitch_dim, energy_dim=energy_dim, **encoder_config)
        if global_encoder_config is not None:
            raise NotImplementedError('MSMCVQGANEmb: the reference encoder (global_encoder_config, ECAPA_TDNN of the '
                                      "reference's tdnn.py) is outside this build; the shipped QS-TTS configuration "
                                      'does not use it')
        self.quantizer = MultiStageQuantizer(n_model_size, list(encoder_config['downsample_scales'])[::-1],
                                             **quantizer_config)
        decoder_config = dict(decoder_config)
        decoder_config['num_mels'] = n_model_size
        self.decoder = HifiGANGenerator(**decoder_config)
        if frame_decoder_config is not None:
            self.frame_decoder = FFTBlocks(d_model=n_model_size, name='frame_decoder', **frame_decoder_config)
        if pred_mel:
            self.mel_predictor = nn.Linear(n_model_size, mel_dim if mel_dim is not None else emb_dim)

    def _decode_frames(self, x, lengths):
        if hasattr(self, 'frame_decoder'):
            x, _ = self.frame_decoder(x, _fft_pos(lengths, x), lengths=lengths)
        return x

    def forward(self, emb, emb_length, pitch=None, energy=None, mel=None, ref=None, window='full'):
        """``window``: None = no waveform (frames only), 'full' = decode every frame, a list of (utterance, start, end)
        frame triples (the reference's convention here, :206-209) or a [B, n] tensor of frame indices = decode those"""
        if self.training:
            hipnorm.advance_seed(emb.device)        # fresh dropout masks for the fused kernels of this step
        enc, content = self.encoder(self.in_linear(emb), emb_length, pitch, energy)
        feats, lens = zip(*enc)
        qs = self.quantizer(enc)
        out = {'encoder_outputs': feats[::-1], 'encoder_lengths': lens[::-1], 'content_representations': content,
               'encoder_indices': qs['quantizer_indices'], 'encoder_diffs': qs['quantizer_diffs'],
               'decoder_diffs': qs['predictor_diffs']}
        dec_in = self._decode_frames(qs['residual_output'], emb_length)
        if hasattr(self, 'mel_predictor'):
            out['mel_outputs'] = self.mel_predictor(dec_in)
        if window is not None:
            if torch.is_tensor(window):
                dec_in = torch.gather(dec_in, 1, window.unsqueeze(-1).expand(-1, -1, dec_in.shape[-1]))
            elif isinstance(window, (list, tuple)):
                dec_in = torch.stack([dec_in[i, s:e] for i, s, e in window], dim=0)
            out['decoder_outputs'] = self.decoder(dec_in.transpose(1, 2)).transpose(1, 2)
        return out

    def analysis(self, emb, emb_length, pitch=None, energy=None):
        enc, content = self.encoder(self.in_linear(emb), emb_length, pitch, energy)
        qs = self.quantizer(enc)
        if self.training:
            feats, lens = zip(*enc)
            return {'encoder_outputs': feats[::-1], 'encoder_lengths': lens[::-1],
                    'encoder_indices': qs['quantizer_indices'], 'encoder_diffs': qs['quantizer_diffs'],
                    'decoder_diffs': qs['predictor_diffs'], 'quantizer_states': qs, 'content_representations': content}
        return qs

    def synthesis(self, quantizer_outputs, quantizer_lengths, ref=None):
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
