#!/usr/bin/env python
"""Inference entry point with the reference's command line (reference infer.py:1-133):

    python infer.py -m checkpoints/model_100000 [-c config.yaml] [-t test.list] [-j batch] [-o out_dir]

Restores the task from a checkpoint (its own configuration unless ``-c`` is given), reads the ``testset`` (or ``dataset``)
section in evaluation mode, runs the task's inference step (analysis-synthesis for ``_mode: train_autoencoder``, text ->
waveform for ``train_predictor``) and writes every ``save_features`` entry ``[name, extension, sample rate]`` per
utterance: ``.wav`` (16-bit PCM, peak-limited), ``.npy``, ``.txt``, ``.dat`` (raw float32).
"""
import argparse
import os
import re
import sys
import wave

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, 'msmc-tts_amd')]
import msmctts_amd  # noqa: E402,F401  (before the first HIP call)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.utils.data import DataLoader, SequentialSampler  # noqa: E402

from msmctts_amd.datasets import build_dataset  # noqa: E402
from msmctts_amd.datasets.base_dataset import feature_normalize  # noqa: E402
from msmctts_amd.tasks import build_task  # noqa: E402
from msmctts_amd.utils.utils import to_model  # noqa: E402


def output_base(checkpoint):
    """<checkpoint dir>/eval-<iteration>"""
    m = re.match(r'.*_([0-9]+)', checkpoint)
    return os.path.join(os.path.dirname(checkpoint), 'eval-%d' % int(m.group(1)) if m else 'eval')


def save_feature(path, feat, fmt, sample_rate):
    if fmt == '.npy':
        np.save(path, feat)
    elif fmt == '.txt':
        np.savetxt(path, feat, fmt='%.6f')
    elif fmt == '.dat':
        feat.astype(np.float32).tofile(path)
    elif fmt == '.wav':
        x = np.asarray(feat, dtype=np.float64).reshape(-1)
        peak = np.abs(x).max() if x.size else 0.0
        x = x / peak if peak > 1 else x
        with wave.open(path, 'wb') as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(int(sample_rate))
            w.writeframes((x * 32767.0).astype('<i2').tobytes())
    else:
        raise ValueError('unsupported output format %r (plots are not part of this package)' % fmt)


def run(task, testset, output_dir, batch_size=1):
    if not hasattr(task.config, 'save_features'):
        raise ValueError('No saved features')
    loader = DataLoader(testset, batch_size=batch_size, num_workers=0, shuffle=False, drop_last=False,
                        sampler=SequentialSampler(testset), collate_fn=getattr(testset, 'collate_fn', None))
    from msmctts_amd.hip import lib as hiplib
    if torch.cuda.is_available():
        task = task.cuda()
    elif not hiplib._host_pointers_ok:
        # Every network of this package computes on the gfx950 kernels; there is no CPU execution path (the reference can
        # synthesise on the CPU, this package cannot): say so here, not from the first encoder call
        raise RuntimeError('infer.py needs an MI355X (gfx950) GPU: msmctts_amd has no CPU execution path')
    task.eval()
    dirs = {}
    for name, _, _ in task.config.save_features:
        dirs[name] = os.path.join(output_dir, name)
        os.makedirs(dirs[name], exist_ok=True)
    for features in loader:
        ids = [testset.id_list[int(i)] for i in features.pop('_id')]          # (collation sorts the batch)
        with torch.no_grad():
            outputs = task(to_model(features))
        for i, uid in enumerate(ids):
            uid = uid if isinstance(uid, str) else '_'.join(uid)
            for name, fmt, sample_rate in task.config.save_features:
                feat = outputs[name][i]
                feat = feat.detach().float().cpu().numpy() if torch.is_tensor(feat) else np.asarray(feat)
                if name in testset.feature_stat:
                    feat = feature_normalize(feat, testset.feature_stat[name], True)
                save_feature('{}/{}{}'.format(dirs[name], uid, fmt), feat, fmt, sample_rate)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-m', '--model', required=True)
    ap.add_argument('-c', '--config', default=None)
    ap.add_argument('-t', '--test_config', default=None)
    ap.add_argument('-j', '--jobs', type=int, default=1)
    ap.add_argument('-o', '--output_dir', default=None)
    ap.add_argument('--debug', action='store_true')
    args = ap.parse_args()
    task = build_task(args.config, mode='debug' if args.debug else 'infer', checkpoint=args.model)
    ds_cfg = task.config.testset if hasattr(task.config, 'testset') else task.config.dataset
    ds_cfg['training'] = False
    if args.test_config is not None:
        ds_cfg['id_list'] = args.test_config
    testset = build_dataset(ds_cfg)
    out = args.output_dir or output_base(args.model)
    os.makedirs(out, exist_ok=True)
    run(task, testset, out, batch_size=args.jobs)


if __name__ == '__main__':
    main()
