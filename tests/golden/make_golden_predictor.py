#!/usr/bin/env python
"""Golden fixture for predictor training (BASELINE config #4), generated from the reference itself.

    python tests/golden/make_golden_predictor.py        # writes tests/golden/small_predictor.npz

A small MultiStagePredictor (reference networks/acoustic_models/multi_stage_predictor.py:9-126) is trained for one step by
the reference's PredictorTrainer.train_step (trainers/msmctts_trainer.py:222-286) against the frozen small autoencoder of
small_state.npz: duration loss + per-stage embedding losses ('mse' and 'triple_sum', i.e. Quantize.compute_triple_loss,
vqgantts/modules.py:86-116).  Stored: predictor state_dict, batch, forward outputs, every loss, every gradient norm.
Data only; no reference source.
"""
import copy
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the import shims)

import torch  # noqa: E402

from msmctts.tasks import build_task  # noqa: E402
from msmctts.trainers.msmctts_trainer import PredictorTrainer  # noqa: E402
from msmctts.trainers.optimizers import build_optimizer  # noqa: E402
from msmctts.utils.config import Config  # noqa: E402

FFT = dict(n_layers=1, n_head=2, d_k=8, d_v=8, d_model=32, d_inner=64, fft_conv1d_kernel=3, fft_conv1d_padding=1,
           dropout=0.0, fused_layernorm=False)
PRED_TASK = {
    '_name': 'MSMCTTS', '_mode': 'train_predictor',
    'predictor': {
        '_name': 'MultiStagePredictor', 'n_symbols': [20, 5, 2], 'n_model_size': 32, 'n_pred_size': 32,
        'n_pred_scale': [4, 1],
        'encoder_config': dict(max_seq_len=32, name='phoneme_side', **FFT),
        'adaptor_config': dict(input_size=32, duration_predictor_filter_size=16, duration_predictor_kernel_size=3,
                               dropout=0.0, fused_layernorm=False),
        'decoder_config': dict(max_seq_len=64, name='mel_side', **FFT),
    },
}
PRED_TRAINER = dict(grad_clip_thresh=10.0, training_methods=['mse', 'triple_sum'], loss_weights=[[1.0, 1.0], [1.0, 1.0]],
                    lambda_dur=1.0)
OPTIM = {'_default': dict(_name='Adam', learning_rate=2e-4, betas=[0.9, 0.98], eps=1e-9, weight_decay=0)}


def main():
    torch.manual_seed(4321)
    cfg = Config({'id': 'golden_predictor', 'task': copy.deepcopy(PRED_TASK), 'trainer': dict(PRED_TRAINER, _name='PredictorTrainer'),
                  'optimizer': copy.deepcopy(OPTIM), 'dataset': copy.deepcopy(G.DATASET),
                  'dataloader': {'batch_size': 3, 'num_workers': 0}})
    task = build_task(cfg, mode='train')
    G.zero_dropout(task)
    task.train()
    out = {}
    for k, v in task.state_dict().items():
        out['state.' + k] = G.npy(v).copy()           # (a copy: the optimizer step below updates in place)
    # frozen small autoencoder with the weights of small_state.npz
    acfg, atask = G.build_small(1234)
    sd = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(HERE, 'small_state.npz')).items()}
    atask.load_state_dict(sd)
    # batch: 3 utterances; phoneme durations sum to the mel lengths of small_batch()
    mel_batch = G.small_batch()
    g = torch.Generator().manual_seed(11)
    text_length = torch.tensor([7, 5, 3], dtype=torch.int64)
    Tt = int(text_length.max())
    text = torch.zeros(3, Tt, 3, dtype=torch.int64)
    dur = torch.zeros(3, Tt, dtype=torch.int64)
    for b, (tl, ml) in enumerate(zip(text_length.tolist(), mel_batch['mel_length'].tolist())):
        text[b, :tl, 0] = torch.randint(1, 20, (tl,), generator=g)
        text[b, :tl, 1] = torch.randint(1, 5, (tl,), generator=g)
        text[b, :tl, 2] = torch.randint(1, 2, (tl,), generator=g)
        cuts = sorted(torch.randperm(ml - 1, generator=g)[:tl - 1].add(1).tolist())
        edges = [0] + cuts + [ml]
        dur[b, :tl] = torch.tensor([edges[i + 1] - edges[i] for i in range(tl)])
    batch = {'text': text, 'text_length': text_length, 'dur': dur, 'mel': mel_batch['mel'], 'mel_length': mel_batch['mel_length']}
    for k, v in batch.items():
        out['batch.' + k] = G.npy(v)

    tr = PredictorTrainer(cfg, task, num_gpus=0, rank=0, **PRED_TRAINER)
    tr.autoencoder = atask.autoencoder
    tr.optimizer = build_optimizer(task, cfg.optimizer)
    # forward outputs of the predictor alone (same weights, train mode, teacher-forced features)
    atask.autoencoder.eval()
    with torch.no_grad():
        qs = atask.autoencoder.analysis(batch['mel'], batch['mel_length'].int())
        fo = task.predictor(text=text, text_length=text_length, dur=dur, feat=qs['quantizer_outputs'],
                            feat_length=qs['quantizer_lengths'])
    for i, f in enumerate(fo['feat']):
        out['fwd.feat.%d' % i] = G.npy(f)
        out['fwd.feat_length.%d' % i] = G.npy(fo['feat_length'][i])
        out['ae.quantizer_outputs.%d' % i] = G.npy(qs['quantizer_outputs'][i])
        out['ae.quantizer_indices.%d' % i] = G.npy(qs['quantizer_indices'][i])
    out['fwd.duration'] = G.npy(fo['duration'])
    snaps = {}
    real_step = tr.optimizer.step

    def spy(names=None):
        key = names[0] if isinstance(names, (list, tuple)) else names
        snaps[key] = {n: p.grad.detach().clone() for n, p in task.named_parameters()
                      if n.startswith(key + '.') and p.grad is not None}
        return real_step(names)
    tr.optimizer.step = spy
    task.zero_grad()
    log = tr.train_step({k: v.clone() for k, v in batch.items()}, 0)
    for k, v in log['loss'].items():
        out['loss.' + k] = np.asarray(float(v), dtype=np.float64)
    names = sorted(snaps['predictor'])
    out['grad_names'] = np.frombuffer(json.dumps(names).encode(), np.uint8)
    out['grad_l2'] = np.asarray([snaps['predictor'][n].double().norm().item() for n in names])
    for k, v in task.state_dict().items():
        if k.endswith(('word_emb.0.weight', 'linear_layer.bias', 'decoders.1.2.bias')):
            out['post.' + k] = G.npy(v)
    np.savez_compressed(os.path.join(HERE, 'small_predictor.npz'), **out)
    print('small_predictor.npz %.1f kB; losses %s' % (os.path.getsize(os.path.join(HERE, 'small_predictor.npz')) / 1024,
                                                      {k: float(v) for k, v in log['loss'].items()}))


if __name__ == '__main__':
    main()
