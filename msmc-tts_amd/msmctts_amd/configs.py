"""The CSMSC MSMC-VQ-GAN configuration as data (values of reference
examples/csmsc/configs/msmc_vq_gan.yaml:7-135), with the BASELINE.json overrides as arguments.

``csmsc_config()`` returns a plain dict accepted by ``Config``; any YAML with the same keys works too.
BASELINE configs: #1 ``downsample_scales=[1], n_heads=1, embedding_sizes=64, batch_size=4``;
#2/#3 ``embedding_sizes=256``; #5 ``in_dim=1024, n_heads=8, embedding_sizes=512``.
"""


def _fft_block_cfg():
    return dict(max_seq_len=2400, n_layers=4, n_head=2, d_k=64, d_v=64, d_inner=1024, fft_conv1d_kernel=3,
                fft_conv1d_padding=1, dropout=0.2, attn_dropout=0.1, fused_layernorm=False)


def csmsc_config(downsample_scales=(1, 4), n_heads=4, embedding_sizes=64, in_dim=80, batch_size=16,
                 warmup_steps=50000, sample_lengths=12000):
    enc = dict(downsample_scales=list(downsample_scales), **_fft_block_cfg())
    return {
        'id': 'msmc_vqgan',
        'task': {
            '_name': 'MSMCTTS', '_mode': 'train_autoencoder',
            'autoencoder': {
                '_name': 'MSMCVQGAN', 'in_dim': in_dim, 'n_model_size': 256,
                'encoder_config': enc,
                'quantizer_config': dict(embedding_sizes=embedding_sizes, embedding_dims=256, n_heads=n_heads,
                                         prior_config=dict(kernel_size=5, dilation_rate=1, n_layers=1), norm=False),
                'frame_decoder_config': _fft_block_cfg(),
                'pred_mel': True,
                'decoder_config': dict(upsample_rates=[6, 5, 5, 2], upsample_kernel_sizes=[12, 11, 11, 4],
                                       upsample_initial_channel=512, resblock_kernel_sizes=[3, 7, 11],
                                       resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]]),
            },
            'discriminator': {
                '_name': 'UnivNetDiscriminator',
                'mrd_config': dict(hop_lengths=[15, 30, 50, 120, 240], hidden_channels=[128, 128, 256, 256, 512],
                                   domain='double', mel_scale=True, sample_rate=24000),
                'mpd_config': dict(periods=[2, 3, 5, 7, 11], channels=16, max_channels=512),
            },
        },
        'save_checkpoint_dir': '', 'pretrain_checkpoint_path': '', 'restore_checkpoint_path': '',
        'resume_training': True, 'training_steps': 800000, 'iters_per_checkpoint': 50000, 'seed': 1234,
        'cudnn': {'enabled': True, 'benchmark': True},
        'trainer': dict(_name='VQGANTrainer', grad_clip_thresh=1.0, warmup_steps=warmup_steps,
                        sample_lengths=sample_lengths, lambda_vq=1, lambda_pr=0.1, lambda_frame=450, lambda_fm=2,
                        lambda_stft=45),
        'optimizer': {'_default': dict(_name='AdamW', learning_rate=2e-4, betas=[0.8, 0.99], eps=1e-8,
                                       weight_decay=0.0)},
        'dataloader': dict(batch_size=batch_size, num_workers=8),
        'dataset': dict(_name='MelDataset', samplerate=24000, feature=['mel', 'wav'], dimension=[in_dim, 1],
                        frameshift=[300, 1], padding_value=[-4, 0], pre_load=False, segment_length=-1),
        'lr_scheduler': dict(_name='ExponentialDecayLRScheduler', warmup_steps=200000, decay_scale=200000,
                             decay_learning_rate=0.5, final_learning_rate=1e-5),
        'distributed': dict(dist_backend='nccl', dist_url='tcp://localhost:54321'),
    }


def am_config(batch_size=64, dropout=0.1):
    """The CSMSC multi-stage predictor configuration as data (values of reference examples/csmsc/configs/msmc_vq_gan_am.yaml:7-124,
    BASELINE.json configuration #4): 600-wide text encoder / frame decoders of six FFT blocks, per-stage predictions of
    256 values at scales [4, 1], 'mse' + 'triple_sum' embedding losses against a frozen MSMC-VQ-GAN autoencoder."""
    fft = lambda name, seq: dict(max_seq_len=seq, n_layers=6, n_head=2, d_k=64, d_v=64, d_model=600, d_inner=1536,
                                 fft_conv1d_kernel=3, fft_conv1d_padding=1, dropout=dropout, name=name, fused_layernorm=False)
    return {
        'id': 'msmc_vq_gan_am',
        'task': {
            '_name': 'MSMCTTS', '_mode': 'train_predictor',
            'predictor': {
                '_name': 'MultiStagePredictor', 'n_symbols': [100, 10, 2], 'n_model_size': 600, 'n_pred_size': 256,
                'n_pred_scale': [4, 1],
                'encoder_config': fft('phoneme_side', 240),
                'adaptor_config': dict(input_size=600, duration_predictor_filter_size=256, duration_predictor_kernel_size=3,
                                       dropout=dropout, fused_layernorm=False),
                'decoder_config': fft('mel_side', 2400),
            },
        },
        'save_checkpoint_dir': '', 'pretrain_checkpoint_path': '', 'restore_checkpoint_path': '',
        'resume_training': True, 'training_steps': 200000, 'iters_per_checkpoint': 50000, 'seed': 1234,
        'cudnn': {'enabled': True, 'benchmark': True},
        'trainer': dict(_name='PredictorTrainer', grad_clip_thresh=10.0, training_methods=['mse', 'triple_sum'],
                        loss_weights=[[1.0, 1.0], [1.0, 1.0]], lambda_dur=1.0),
        'optimizer': {'_default': dict(_name='Adam', learning_rate=2e-4, betas=[0.9, 0.98], eps=1e-9, weight_decay=0)},
        'dataloader': dict(batch_size=batch_size, num_workers=8),
        'dataset': dict(_name='TTSDataset', samplerate=24000, feature=['text', 'dur', 'mel'], dimension=[3, 1, 80],
                        padding_value=[0, 0, -4], frameshift=[None, None, 200], pre_load=True, segment_length=-1),
        'lr_scheduler': dict(_name='ExponentialDecayLRScheduler', warmup_steps=20000, decay_scale=20000,
                             decay_learning_rate=0.5, final_learning_rate=1e-6),
        'distributed': dict(dist_backend='nccl', dist_url='tcp://localhost:54321'),
    }


# BASELINE.json ``configs`` as overrides of ``csmsc_config`` (bench.py --config N, tests/test_gpu_fullsize.py).
# #3 "LJSpeech msmc_vq_gan.yaml": the reference ships no such file (its LJSpeech YAMLs are v1-era and name classes that do
# not exist); it is the CSMSC architecture with LJSpeech audio parameters -- 24 kHz, hop 300,
# identical to CSMSC (reference examples/ljspeech/voc/configs/hifigan.yaml:61-68) -- and 256 codewords, i.e. the SAME model and
# batch contract as #2, run data-parallel on the 8 GPUs of a node (python -m torch.distributed.run --nproc-per-node 8
# bench.py --gpus 8 --config 3).  #4 (predictor training against a frozen autoencoder of configuration #2's architecture):
# ``am_config`` above, ``bench.py --config 4``.
BASELINE_CONFIGS = {
    1: dict(name='CSMSC msmc_vq_gan (plumbing case)', model=dict(downsample_scales=(1,), n_heads=1, embedding_sizes=64),
            per_gpu_batch=4, gpus=1),
    2: dict(name='CSMSC msmc_vq_gan', model=dict(n_heads=4, embedding_sizes=256), per_gpu_batch=16, gpus=1),
    3: dict(name='LJSpeech msmc_vq_gan (CSMSC architecture, LJSpeech audio parameters = CSMSC\'s)',
            model=dict(n_heads=4, embedding_sizes=256), per_gpu_batch=16, gpus=8),
    4: dict(name='CSMSC msmc_vq_gan_am (multi-stage predictor against a frozen MSMC-VQ-GAN autoencoder)',
            model=dict(n_heads=4, embedding_sizes=256), per_gpu_batch=64, gpus=1),
    5: dict(name='QS-TTS msmc_vq_gan HuBERT-feature stress', model=dict(in_dim=1024, n_heads=8, embedding_sizes=512),
            per_gpu_batch=16, gpus=8),
}


def baseline_config(index, **overrides):
    """the ``csmsc_config`` dictionary of BASELINE.json configuration ``index`` (1, 2, 3 or 5)"""
    preset = BASELINE_CONFIGS[index]
    kw = dict(preset['model'], batch_size=preset['per_gpu_batch'])
    kw.update(overrides)
    return csmsc_config(**kw)
