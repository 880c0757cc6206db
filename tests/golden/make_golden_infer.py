#!/usr/bin/env python
"""Golden fixture for the inference glue, generated from the reference itself.

    python tests/golden/make_golden_infer.py        # writes tests/golden/small_infer.npz

The reference's MSMCTTS task (msmctts/tasks/msmc_tts.py:87-151) in evaluation mode: analysis-synthesis of the small
autoencoder (small_state.npz) on small_batch(), and text -> waveform through the small predictor (small_predictor.npz
state) + the same autoencoder's ``synthesis`` with teacher durations.  Stored: the waveforms (every 3rd sample of the first
3 000 per utterance, plus length / mean / mean-abs digests) and the last-stage embedding.  Data only; no reference source.
"""
import copy
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the import shims)
import make_golden_predictor as P  # noqa: E402

import torch  # noqa: E402

from msmctts.tasks import build_task  # noqa: E402
from msmctts.utils.config import Config  # noqa: E402


def digest(w):
    w = w.detach().double().reshape(-1)
    return np.concatenate([[w.numel(), w.mean().item(), w.abs().mean().item()], w[:3000:3].numpy()])


def main():
    out = {}
    acfg, atask = G.build_small(1234)
    atask.load_state_dict({k: torch.from_numpy(v) for k, v in np.load(os.path.join(HERE, 'small_state.npz')).items()})
    atask.eval()
    batch = G.small_batch()
    with torch.no_grad():
        res = atask.infer_step({'mel': batch['mel'], 'mel_length': batch['mel_length']}, mode='train_autoencoder')
    for i, w in enumerate(res['wav']):
        out['ae.wav.%d' % i] = digest(w)
    z = np.load(os.path.join(HERE, 'small_predictor.npz'))
    cfg = Config({'id': 'golden_infer', 'task': copy.deepcopy(P.PRED_TASK), 'dataset': copy.deepcopy(G.DATASET)})
    task = build_task(cfg, mode='infer')
    task.load_state_dict({k[len('state.'):]: torch.from_numpy(z[k]) for k in z.files if k.startswith('state.')})
    task.eval()
    task.autoencoder, task.load_modules = atask.autoencoder, True
    feed = {k: torch.from_numpy(z['batch.' + k]) for k in ('text', 'text_length', 'dur')}
    with torch.no_grad():
        res = task.infer_step(feed, mode='train_predictor')
    for i, w in enumerate(res['wav']):
        out['tts.wav.%d' % i] = digest(w)
    out['tts.embedding'] = G.npy(res['embedding']).copy()
    out['tts.duration'] = G.npy(res['duration']).copy()
    path = os.path.join(HERE, 'small_infer.npz')
    np.savez_compressed(path, **out)
    print('small_infer.npz %.1f kB; wav lengths %s / %s' % (os.path.getsize(path) / 1024,
          [int(out['ae.wav.%d' % i][0]) for i in range(3)], [int(out['tts.wav.%d' % i][0]) for i in range(3)]))


if __name__ == '__main__':
    main()
