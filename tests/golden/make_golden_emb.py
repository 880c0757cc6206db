#!/usr/bin/env python
"""Golden fixture for the QS-TTS synthesiser, generated from the reference itself.

    python tests/golden/make_golden_emb.py        # writes tests/golden/small_emb.npz

A small MSMCVQGANEmb (reference networks/vqgantts/msmc_vqgan_emb.py:123-291; the module it imports under the name
msmc_vqgan_speech is aliased to msmc_vqgan.py by _ref_shims.py) with the pitch / energy side encoder on:
training-mode forward over windows (every dictionary entry, the gradient of a scalar of them with respect to the input
embeddings), training-mode analysis, evaluation-mode analysis -> synthesis.  Stored: state_dict, batch, outputs.
Data only; no reference source.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the import shims)

import torch  # noqa: E402

from msmctts.networks.vqgantts.msmc_vqgan_emb import MSMCVQGANEmb  # noqa: E402

FFT = dict(max_seq_len=64, n_layers=1, n_head=2, d_k=8, d_v=8, d_inner=64, fft_conv1d_kernel=3, fft_conv1d_padding=1,
           dropout=0.0, attn_dropout=0.0, fused_layernorm=False)
EMB_CFG = dict(emb_dim=24, n_model_size=32, pitch_dim=1, energy_dim=1,
               encoder_config=dict(downsample_scales=[1, 4], **FFT),
               quantizer_config=dict(embedding_sizes=16, embedding_dims=32, n_heads=4,
                                     prior_config=dict(kernel_size=5, dilation_rate=1, n_layers=1), norm=False, dropout=0.0),
               frame_decoder_config=dict(FFT), pred_mel=True, mel_dim=20,
               decoder_config=dict(upsample_rates=[5, 4, 2], upsample_kernel_sizes=[11, 8, 4], upsample_initial_channel=32,
                                   resblock_kernel_sizes=[3, 7], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5]]))
WINDOWS = [(0, 3, 11), (1, 0, 8), (2, 1, 9)]          # (utterance, first frame, end frame): the reference's triples


def main():
    torch.manual_seed(2468)
    import copy
    m = MSMCVQGANEmb(**copy.deepcopy(EMB_CFG))
    G.zero_dropout(m)
    out = {'cfg': np.frombuffer(json.dumps(EMB_CFG).encode(), dtype=np.uint8),
           'windows': np.asarray(WINDOWS, dtype=np.int64)}
    for k, v in m.state_dict().items():
        out['state.' + k] = G.npy(v).copy()
    g = torch.Generator().manual_seed(5)
    lengths = torch.tensor([24, 17, 9], dtype=torch.int64)
    emb = torch.randn(3, 24, 24, generator=g)
    pitch, energy = torch.randn(3, 24, 1, generator=g), torch.rand(3, 24, 1, generator=g)
    for i, n in enumerate(lengths.tolist()):
        emb[i, n:], pitch[i, n:], energy[i, n:] = 0.0, 0.0, 0.0
    for k, v in (('emb', emb), ('emb_length', lengths), ('pitch', pitch), ('energy', energy)):
        out['batch.' + k] = G.npy(v)

    def put(prefix, d):
        for k, v in d.items():
            if torch.is_tensor(v):
                out['%s.%s' % (prefix, k)] = G.npy(v)
            elif isinstance(v, (tuple, list)):
                for i, t in enumerate(v):
                    if torch.is_tensor(t):
                        out['%s.%s.%d' % (prefix, k, i)] = G.npy(t)
            elif isinstance(v, dict):
                put('%s.%s' % (prefix, k), v)

    # training-mode forward over windows (the codebooks take one EMA step) + input gradient of a scalar of the outputs
    m.train()
    e = emb.clone().requires_grad_(True)
    o = m(e, lengths, pitch, energy, window=WINDOWS)
    put('train', o)
    scalar = (o['decoder_outputs'].pow(2).mean() + o['mel_outputs'].mean() + sum(d.mean() for d in o['encoder_diffs'])
              + o['decoder_diffs']['total_loss'] + o['content_representations'].mean())
    scalar.backward()
    out['train.scalar'] = G.npy(scalar)
    out['train.grad_emb'] = G.npy(e.grad)
    for k, v in m.state_dict().items():
        if 'quantizer.quantizer' in k:
            out['after.' + k] = G.npy(v).copy()
    # training-mode analysis (second EMA step), then evaluation-mode analysis -> synthesis with the updated codebooks
    a = m.analysis(emb, lengths, pitch, energy)
    put('train_analysis', {k: v for k, v in a.items() if k != 'quantizer_states'})
    m.eval()
    with torch.no_grad():
        qs = m.analysis(emb, lengths, pitch, energy)
        put('eval_analysis', qs)
        wav = m.synthesis(qs, qs['quantizer_lengths'])
        out['eval.wav'] = G.npy(wav)
        wav2 = m.synthesis([q for q in qs['quantizer_outputs']], qs['quantizer_lengths'])     # from the quantised sequences
        out['eval.wav_from_sequences'] = G.npy(wav2)
        full = m(emb, lengths, pitch, energy)                                                  # window='full'
        out['eval.full.decoder_outputs'] = G.npy(full['decoder_outputs'])
    np.savez_compressed(os.path.join(HERE, 'small_emb.npz'), **out)
    print('wrote small_emb.npz: %d arrays' % len(out))


if __name__ == '__main__':
    main()
