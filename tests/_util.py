"""Shared helpers for the test-suite (fixture loading, tolerances)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def json_field(arr):
    return json.loads(bytes(arr.tolist()).decode())


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def small_task_cfg():
    """The SURVEY appendix-C small model; mirrors tests/golden/make_golden.py::SMALL_TASK (data, not code)."""
    fft = dict(max_seq_len=64, n_layers=1, n_head=2, d_k=8, d_v=8, d_inner=64, fft_conv1d_kernel=3,
               fft_conv1d_padding=1, dropout=0.0, attn_dropout=0.0, fused_layernorm=False)
    return {
        'autoencoder': {
            '_name': 'MSMCVQGAN', 'in_dim': 80, 'n_model_size': 32,
            'encoder_config': dict(downsample_scales=[1, 4], **fft),
            'quantizer_config': dict(embedding_sizes=16, embedding_dims=32, n_heads=4,
                                     prior_config=dict(kernel_size=5, dilation_rate=1, n_layers=1),
                                     norm=False, dropout=0.0),
            'frame_decoder_config': dict(fft),
            'pred_mel': True,
            'decoder_config': dict(upsample_rates=[6, 5, 5, 2], upsample_kernel_sizes=[12, 11, 11, 4],
                                   upsample_initial_channel=32, resblock_kernel_sizes=[3, 7, 11],
                                   resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]]),
        },
        'discriminator': {
            '_name': 'UnivNetDiscriminator',
            'mrd_config': dict(hop_lengths=[15, 60], hidden_channels=[32, 32], domain='double',
                               mel_scale=True, sample_rate=24000),
            'mpd_config': dict(periods=[2, 3], channels=4, max_channels=16),
        },
    }


def small_predictor_cfg():
    """The predictor of tests/golden/make_golden_predictor.py::PRED_TASK (data, not code); every nn.Dropout was zeroed when
    the fixture was generated, hence attn_dropout=0."""
    fft = dict(n_layers=1, n_head=2, d_k=8, d_v=8, d_model=32, d_inner=64, fft_conv1d_kernel=3, fft_conv1d_padding=1,
               dropout=0.0, attn_dropout=0.0, fused_layernorm=False)
    return {'_name': 'MultiStagePredictor', 'n_symbols': [20, 5, 2], 'n_model_size': 32, 'n_pred_size': 32,
            'n_pred_scale': [4, 1],
            'encoder_config': dict(max_seq_len=32, name='phoneme_side', **fft),
            'adaptor_config': dict(input_size=32, duration_predictor_filter_size=16, duration_predictor_kernel_size=3,
                                   dropout=0.0, fused_layernorm=False),
            'decoder_config': dict(max_seq_len=64, name='mel_side', **fft)}


PREDICTOR_TRAINER = dict(grad_clip_thresh=10.0, training_methods=['mse', 'triple_sum'],
                         loss_weights=[[1.0, 1.0], [1.0, 1.0]], lambda_dur=1.0)
SMALL_TRAINER = dict(grad_clip_thresh=1.0, warmup_steps=5, sample_lengths=2400, lambda_vq=1, lambda_pr=0.1,
                     lambda_frame=450, lambda_fm=2, lambda_stft=45)


def fmap_digest(tensor):
    f = tensor.detach().reshape(-1).double().cpu()
    return np.concatenate([[f.mean().item(), f.abs().mean().item(), f.std().item(), float(f.numel())],
                           f[::7][:512].numpy()]).astype(np.float64)
