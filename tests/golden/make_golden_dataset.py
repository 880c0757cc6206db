#!/usr/bin/env python
"""Golden fixture for the data path, generated from the reference itself.

    python tests/golden/make_golden_dataset.py        # writes tests/golden/dataset_cases.npz

Five synthetic utterances are written as feature files (.npy mel [T, 80] in C and Fortran order, .npy wav [T*hop + r, 1],
a per-utterance scalar from a "book" .list file) and read back by the reference's MelDataset
(msmctts/datasets/mel_dataset.py:9-66 over base_dataset.py:24-286): training mode with random 40-frame segments, read
from the files (pre_load=False) and from pre-loaded arrays, and evaluation mode (whole utterances); each case is collated
into a batch.  Stored: the file contents (inputs), the constructor arguments, and every batch tensor (expected outputs).
Data only; no reference source.
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402,F401  (installs the import shims)

from msmctts.datasets.mel_dataset import MelDataset  # noqa: E402

LENGTHS = [57, 93, 41, 120, 64]
HOP = 30                              # (a tenth of the real hop keeps the fixture small)
EXTRA = [0, 7, 15, 29, 3]                 # wav samples beyond T * HOP (alignment must trim them)


def write_corpus(root):
    rng = np.random.default_rng(99)
    os.makedirs(os.path.join(root, 'mel'))
    os.makedirs(os.path.join(root, 'wav'))
    files = {}
    ids = ['utt%02d' % i for i in range(len(LENGTHS))]
    for i, (uid, T, r) in enumerate(zip(ids, LENGTHS, EXTRA)):
        mel = rng.standard_normal((T, 80)).astype(np.float32)
        wav = rng.uniform(-1, 1, (T * HOP + r, 1)).astype(np.float32)
        np.save(os.path.join(root, 'mel', uid + '.npy'), np.asfortranarray(mel) if i % 2 else mel)
        np.save(os.path.join(root, 'wav', uid + '.npy'), wav)
        files['file.mel.' + uid], files['file.wav.' + uid] = mel, wav
    with open(os.path.join(root, 'id.list'), 'w') as f:
        f.write('\n'.join(ids) + '\n')
    with open(os.path.join(root, 'spk.list'), 'w') as f:
        f.write('\n'.join('%s|%d' % (uid, i % 3) for i, uid in enumerate(ids)) + '\n')
    return ids, files


def main():
    out = {}
    with tempfile.TemporaryDirectory() as root:
        ids, files = write_corpus(root)
        out.update(files)
        common = dict(id_list=os.path.join(root, 'id.list'), feature=['mel', 'wav', 'spk'], samplerate=24000,
                      dimension=[80, 1, 1], frameshift=[HOP, 1, None],
                      feature_path=[os.path.join(root, 'mel', '{}.npy'), os.path.join(root, 'wav', '{}.npy'),
                                    os.path.join(root, 'spk.list')],
                      padding_value=[-4.0, 0.0, 0.0], seed=4321)
        cases = {'train_files': dict(segment_length=40 * HOP, pre_load=False, training=True),
                 'train_preload': dict(segment_length=40 * HOP, pre_load=True, training=True),
                 'eval': dict(segment_length=-1, pre_load=False, training=False)}
        meta = {'ids': ids, 'cases': {}}
        for name, extra in cases.items():
            ds = MelDataset(**dict(common, **extra))
            order = [list(x) for x in ds.id_list]
            items = [ds[i] for i in (3, 0, 4, 1, 2, 7)]             # (7 wraps around: index % len)
            batch = ds.collate_fn(items)
            meta['cases'][name] = dict(extra, order=order, length=len(ds), keys=sorted(batch.keys()))
            for k, v in batch.items():
                out['%s.%s' % (name, k)] = np.asarray(v)
    out['meta'] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    path = os.path.join(HERE, 'dataset_cases.npz')
    np.savez_compressed(path, **out)
    print('dataset_cases.npz %.1f kB; cases %s' % (os.path.getsize(path) / 1024, {k: v['keys'] for k, v in meta['cases'].items()}))


if __name__ == '__main__':
    main()
