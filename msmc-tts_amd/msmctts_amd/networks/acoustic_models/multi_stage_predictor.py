"""Multi-stage predictor (text -> per-stage VQ embeddings), drop-in for reference
msmctts/networks/acoustic_models/multi_stage_predictor.py:9-126 (BASELINE config #4).

Same constructor arguments, sub-module names (``word_emb``, ``encoder``, ``upsampler.duration_predictor``,
``downsamplers``, ``decoders.<i>.{0,1,2}``: the checkpoint keys) and output dictionary.  The FFT-block stacks run on the
gfx950 kernels like the autoencoder's (transformer.py), and so do the frame-rate layers around them -- the two
down-sampling convolutions (600 -> 600 over every frame of the batch, kernel 2 s + 1) and the per-stage input / output
projections, one ConvBank of plain layers (on stock operators these were a fifth of the step: 7 of 37.6 ms at B = 64,
profiles/r05_config4_kernel_stats.txt).  The length regulator expands all utterances in one gather instead of the
reference's per-utterance ``repeat_interleave`` loop (transformer.py:459-478) -- same result.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...hip.convnet import ConvBank, ConvLayer, hip_conv
from ...utils.utils import get_mask_from_lengths
from .transformer import FFTBlocks, LengthRegulator, _interpreter_bound


def _positions(lengths, device, width=None):
    width = int(lengths.max()) if width is None else width
    pos = torch.arange(1, width + 1, device=device).unsqueeze(0).repeat(lengths.shape[0], 1)
    return pos.masked_fill(get_mask_from_lengths(lengths.to(device), width), 0)


class MultiStagePredictor(nn.Module):
    def __init__(self, n_symbols, n_model_size, n_pred_size, n_pred_scale, encoder_config, adaptor_config, decoder_config):
        super().__init__()
        self.n_pred_scale, self.n_symbols = list(n_pred_scale), n_symbols
        if isinstance(n_symbols, (tuple, list)):
            self.word_emb = nn.ModuleList([nn.Embedding(n, n_model_size, padding_idx=0) for n in n_symbols])
        else:
            self.word_emb = nn.Embedding(n_symbols, n_model_size, padding_idx=0)
        self.encoder = FFTBlocks(**encoder_config)
        self.upsampler = LengthRegulator(**adaptor_config)
        self.downsamplers = nn.ModuleList([nn.Conv1d(n_model_size, n_model_size, s * 2 + 1, padding=s)
                                           for s in self.n_pred_scale[::-1]])
        self.decoders = nn.ModuleList([
            nn.ModuleList([nn.Linear(n_model_size * 2 + n_pred_size if i > 0 else n_model_size, n_model_size),
                           FFTBlocks(**decoder_config), nn.Linear(n_model_size, n_pred_size)])
            for i in range(len(self.n_pred_scale))])
        self.hip_dtype = torch.float32        # compute dtype of the frame-rate glue layers (trainer: bfloat16 in bf16 runs)
        self._bank = None

    def _hip_ready(self, dev):
        """(build and) refresh the kernel-layout weights of the down-sampling convolutions and the stage projections"""
        if not (dev.type == 'cuda' or _interpreter_bound()):
            return False
        if self._bank is None:
            self._layers = {id(m): ConvLayer(m, 'conv', (1, m.kernel_size[0]), (1, 1), (1, 1), (0, m.padding[0]), plain=True)
                            for m in self.downsamplers}
            for dec in self.decoders:
                self._layers.update({id(m): ConvLayer(m, 'conv', (1, 1), plain=True) for m in (dec[0], dec[2])})
            self._bank = ConvBank(list(self._layers.values()))
        self._bank.prepare(self.hip_dtype)
        return True

    def _glue(self, module, x, hip):
        """``module`` (a down-sampling Conv1d or a stage's Linear) on channels-last x (B, T, C)"""
        if hip:
            return hip_conv(self._bank, self._layers[id(module)], x.to(self.hip_dtype).contiguous().unsqueeze(1)).squeeze(1)
        if isinstance(module, nn.Conv1d):
            return module(x.transpose(1, 2)).transpose(1, 2)
        return module(x)

    def forward(self, text, text_length, dur=None, feat=None, feat_length=None, frames=None):
        """``frames`` (not in the reference): the batch's frame count as a host integer (>= the longest utterance, e.g. the
        padded mel width).  With it no length is read back from the device -- the expansion and the per-stage position tables
        take their widths from it and from the tensors' shapes -- so the call can be captured into a hipGraph; rows past an
        utterance's length are padding either way."""
        output, duration = self.encode(text, text_length, dur, frames)
        if feat_length is None:
            total = duration.sum(-1).long()
            feat_length = []
            for scale in self.n_pred_scale[::-1]:
                total = torch.ceil(total / scale).long()
                feat_length.append(total)
            feat_length = feat_length[::-1]
        return {'feat': self.decode(output, feat, feat_length, static=frames is not None), 'feat_length': feat_length, 'text_length': text_length,
                'duration': duration}

    def encode(self, text, text_length, dur=None, frames=None):
        if isinstance(self.n_symbols, (tuple, list)):
            output = sum(emb(text[..., i].long()) for i, emb in enumerate(self.word_emb))
        else:
            output = self.word_emb(text.long())
        output, text_mask = self.encoder(output, _positions(text_length, text.device, text.shape[1]))
        output, _, duration = self.upsampler(output.float(), text_mask, target=dur, alpha=1.0, width=frames)
        return output, duration

    def decode(self, text_embedding, feat=None, feat_lengths=None, static=False):
        hip = self._hip_ready(text_embedding.device)
        downsampled = []
        for conv, scale in zip(self.downsamplers, self.n_pred_scale[::-1]):
            text_embedding = self._glue(conv, text_embedding, hip)
            if scale != 1:
                text_embedding = F.avg_pool1d(text_embedding.transpose(1, 2), kernel_size=scale, stride=scale,
                                              ceil_mode=True).transpose(1, 2)
            downsampled.append(text_embedding)
        downsampled = downsampled[::-1]
        predictions, output = [], None
        for i, decoder in enumerate(self.decoders):
            emb = downsampled[i]
            pos = _positions(feat_lengths[i], emb.device, emb.shape[1] if static else None)
            if i > 0:
                scale = self.n_pred_scale[i - 1]
                pre = feat[i - 1] if feat is not None else predictions[-1]
                pre = torch.cat((output, pre.to(output.dtype)), dim=2)
                pre = torch.repeat_interleave(pre, scale, dim=1)[:, :emb.shape[1]]
                output = torch.cat((emb, pre), dim=2)
            else:
                output = emb
            output = self._glue(decoder[0], output, hip)
            output, _ = decoder[1](output[:, :pos.shape[1]], pos)
            output = output.float()
            prediction = self._glue(decoder[2], output, hip).float()
            if not self.training and hasattr(self, 'quantizers'):
                q = self.quantizers[i]
                prediction = (q.quantize if hasattr(q, 'quantize') else q)(prediction)[0]
            predictions.append(prediction)
        return predictions
