"""Data path: the product's MelDataset against batches produced by the reference's own MelDataset
on the same files (tests/golden/dataset_cases.npz, make_golden_dataset.py), the file readers, and the device loader."""
import os
import sys
import wave
import zipfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd')]
from _util import json_field, load_npz  # noqa: E402


def _corpus(root, z, meta):
    os.makedirs(os.path.join(root, 'mel'))
    os.makedirs(os.path.join(root, 'wav'))
    for i, uid in enumerate(meta['ids']):
        mel = z['file.mel.' + uid]
        np.save(os.path.join(root, 'mel', uid + '.npy'), np.asfortranarray(mel) if i % 2 else mel)
        np.save(os.path.join(root, 'wav', uid + '.npy'), z['file.wav.' + uid])
    with open(os.path.join(root, 'id.list'), 'w') as f:
        f.write('\n'.join(meta['ids']) + '\n')
    with open(os.path.join(root, 'spk.list'), 'w') as f:
        f.write('\n'.join('%s|%d' % (u, i % 3) for i, u in enumerate(meta['ids'])) + '\n')
    return dict(id_list=os.path.join(root, 'id.list'), feature=['mel', 'wav', 'spk'], samplerate=24000,
                dimension=[80, 1, 1], frameshift=[30, 1, None],
                feature_path=[os.path.join(root, 'mel', '{}.npy'), os.path.join(root, 'wav', '{}.npy'),
                              os.path.join(root, 'spk.list')], padding_value=[-4.0, 0.0, 0.0], seed=4321)


@pytest.mark.parametrize('case', ['train_files', 'train_preload', 'eval'])
def test_mel_dataset_matches_reference_batches(case, tmp_path):
    """utterance order after the seeded shuffle, random 40-frame windows (read from the files or from pre-loaded arrays,
    C- and Fortran-ordered .npy), trimming of unaligned waveforms, the book feature, collation: bit-identical"""
    from msmctts_amd.datasets.mel_dataset import MelDataset
    z = load_npz('dataset_cases.npz')
    meta = json_field(z['meta'])
    c = meta['cases'][case]
    common = _corpus(str(tmp_path), z, meta)
    ds = MelDataset(**dict(common, segment_length=c['segment_length'], pre_load=c['pre_load'], training=c['training']))
    assert [list(x) for x in ds.id_list] == c['order'] and len(ds) == c['length']
    batch = ds.collate_fn([ds[i] for i in (3, 0, 4, 1, 2, 7)])
    assert sorted(batch.keys()) == c['keys']
    for k, v in batch.items():
        want, got = z['%s.%s' % (case, k)], np.asarray(v)
        assert got.shape == want.shape and got.dtype == want.dtype, (k, got.shape, want.shape, got.dtype, want.dtype)
        assert np.array_equal(got, want), k
    # the batch contract of VQGANTrainer.train_step
    assert torch.equal(batch['mel_length'], torch.sort(batch['mel_length'], descending=True)[0])
    assert batch['mel'].shape[1] == int(batch['mel_length'].max()) and torch.equal(batch['wav_length'], batch['mel_length'] * 30)


def test_readers_windows_zip_members_wav_and_raw(tmp_path):
    from msmctts_amd.datasets import readers
    rng = np.random.default_rng(0)
    a = rng.standard_normal((50, 7)).astype(np.float32)
    np.save(tmp_path / 'c.npy', a)
    np.save(tmp_path / 'f.npy', np.asfortranarray(a))
    np.save(tmp_path / 'v.npy', a[:, 0].copy())
    for name in ('c.npy', 'f.npy'):
        p = str(tmp_path / name)
        assert readers.read_npy(p, shape_only=True) == (50, 7)
        assert np.array_equal(readers.read_npy(p), a)
        assert np.array_equal(readers.read_npy(p, 13, 9), a[13:22])
        assert np.array_equal(readers.read_npy(p, 45, 20), a[45:])              # window clipped at the end
    assert readers.read_npy(str(tmp_path / 'v.npy'), 3, 5).shape == (5, 1)      # a windowed vector reads as [n, 1]
    with pytest.raises(ValueError):
        readers.read_npy(str(tmp_path / 'c.npy'), 50, 4)
    with zipfile.ZipFile(tmp_path / 'feats.zip', 'w') as zf:
        zf.write(tmp_path / 'c.npy', 'mel/utt.npy')
    assert np.array_equal(readers.read_npy(str(tmp_path / 'feats.zip') + ':mel/utt.npy', 4, 6), a[4:10])
    pcm = (rng.uniform(-1, 1, 4000) * 32767).astype('<i2')
    with wave.open(str(tmp_path / 'x.wav'), 'wb') as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(24000)
        w.writeframes(pcm.tobytes())
    assert readers.read_wav(str(tmp_path / 'x.wav'), shape_only=True) == (4000, 1)
    x, rate = readers.read_wav(str(tmp_path / 'x.wav'), 100, 250)
    assert rate == 24000 and x.shape == (250, 1) and np.allclose(x[:, 0], pcm[100:350] / 32768.0)
    a.tofile(tmp_path / 'm.mgc')
    assert np.array_equal(readers.read_raw_float32(str(tmp_path / 'm.mgc'), 7), a)


def test_tts_dataset_durations_in_seconds_become_frames(tmp_path):
    """predictor batches: durations given in seconds (recognised, as in the reference, by more than 100 frames per unit of
    duration: hop 200 at 24 kHz is 120) are converted with the rounding error carried forward and reconciled with the mel
    length; the batch is sorted by decreasing text length"""
    from msmctts_amd.datasets.tts_dataset import TTSDataset
    rng = np.random.default_rng(1)
    os.makedirs(tmp_path / 'mel')
    rows_text, rows_dur = [], []
    for uid, n_ph, T in (('a', 5, 60), ('b', 8, 96)):
        np.save(tmp_path / 'mel' / (uid + '.npy'), rng.standard_normal((T, 80)).astype(np.float32))
        cuts = np.sort(rng.choice(np.arange(1, T), n_ph - 1, replace=False))
        frames = np.diff(np.concatenate(([0], cuts, [T])))
        rows_text.append('%s|%s' % (uid, ' '.join(str(v) for v in rng.integers(1, 20, n_ph))))
        rows_dur.append('%s|%s' % (uid, ' '.join('%.6f' % (f * 200 / 24000.0) for f in frames)))
    (tmp_path / 'id.list').write_text('a\nb\n')
    (tmp_path / 'text.list').write_text('\n'.join(rows_text) + '\n')
    (tmp_path / 'dur.list').write_text('\n'.join(rows_dur) + '\n')
    ds = TTSDataset(id_list=str(tmp_path / 'id.list'), feature=['text', 'dur', 'mel'], samplerate=24000,
                    dimension=[1, 1, 80], frameshift=[None, None, 200],
                    feature_path=[str(tmp_path / 'text.list'), str(tmp_path / 'dur.list'), str(tmp_path / 'mel' / '{}.npy')],
                    padding_value=[0, 0, -4.0], training=False)
    batch = ds.collate_fn([ds[0], ds[1]])
    assert batch['text_length'].tolist() == [8, 5] and batch['text'].shape == (2, 8)
    assert batch['dur'].sum(1).tolist() == batch['mel_length'].tolist() == [96.0, 60.0]
    assert torch.equal(batch['dur'], batch['dur'].round())


def test_device_loader_static_shapes_and_host_lengths():
    from msmctts_amd.datasets import DeviceLoader
    batches = [dict(mel=torch.randn(3, T, 80), wav=torch.randn(3, T * 30, 1), mel_length=torch.tensor([T, T - 2, 5]),
                    wav_length=torch.tensor([T, T - 2, 5]) * 30) for T in (37, 40, 12)]
    out = list(DeviceLoader(batches, 'cpu', pad_frames=40, hop=30, mel_pad=-4.0))
    assert len(out) == 3
    for src, b in zip(batches, out):
        T = src['mel'].shape[1]
        assert b['mel'].shape == (3, 40, 80) and b['wav'].shape == (3, 1200, 1)
        assert torch.equal(b['mel'][:, :T], src['mel']) and bool((b['mel'][:, T:] == -4.0).all()) and bool((b['wav'][:, T * 30:] == 0).all())
        assert b['mel_length_host'] == src['mel_length'].tolist()
    with pytest.raises(AssertionError):
        list(DeviceLoader([batches[1]], 'cpu', pad_frames=39, hop=30))


@pytest.mark.gpu
def test_device_loader_uploads_ahead_on_a_side_stream():
    from msmctts_amd.datasets import DeviceLoader
    batches = [dict(mel=torch.randn(4, 50, 80), wav=torch.randn(4, 1500, 1), mel_length=torch.tensor([50, 40, 30, 20]))
               for _ in range(4)]
    out = list(DeviceLoader(batches, 'cuda:0'))
    torch.cuda.synchronize()
    for src, b in zip(batches, out):
        assert b['mel'].is_cuda and torch.equal(b['mel'].cpu(), src['mel']) and torch.equal(b['wav'].cpu(), src['wav'])
        assert b['mel_length_host'] == [50, 40, 30, 20]


def test_training_loop_from_feature_files_to_checkpoint(tmp_path):
    """config.dataset -> MelDataset -> DataLoader -> DeviceLoader -> BaseTrainer.train() on the kernel interpreter: two
    iterations from .npy files, checkpoint written, and a second trainer resumes from it (reference base_trainer.py:29-142)"""
    import subprocess
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'tests', 'emu')])
    from msmctts_amd.hip import lib
    lib.use_library_for_tests(os.path.join(ROOT, 'tests', 'emu', 'libmsmc_emu.so'))
    import _parity
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.utils.config import ConfigItem
    root = str(tmp_path)
    os.makedirs(root + '/mel')
    os.makedirs(root + '/wav')
    rng = np.random.default_rng(0)
    ids = []
    for i, T in enumerate((30, 41, 26, 35)):
        ids.append('u%d' % i)
        np.save(root + '/mel/u%d.npy' % i, rng.standard_normal((T, 80)).astype(np.float32))
        np.save(root + '/wav/u%d.npy' % i, rng.uniform(-1, 1, (T * 300, 1)).astype(np.float32))
    with open(root + '/id.list', 'w') as f:
        f.write('\n'.join(ids) + '\n')

    def trainer():
        cfg, task = _parity.build_small('cpu')
        cfg.dataset = ConfigItem(dict(_name='MelDataset', id_list=root + '/id.list', feature=['mel', 'wav'], samplerate=24000,
                                      dimension=[80, 1], frameshift=[300, 1],
                                      feature_path=[root + '/mel/{}.npy', root + '/wav/{}.npy'], padding_value=[-4.0, 0.0],
                                      segment_length=24 * 300, pre_load=False))
        cfg.dataloader = ConfigItem(dict(batch_size=3, num_workers=0))
        cfg.save_checkpoint_dir, cfg.iters_per_checkpoint = root + '/ckpt', 1
        return cfg, build_trainer(cfg, task, num_gpus=0, rank=0)

    cfg, tr = trainer()
    cfg.training_steps = 1
    seen = []
    assert tr.train(logger=lambda i, log: seen.append((i, {k: float(v) for k, v in log['loss'].items()}))) == 1
    assert [i for i, _ in seen] == [0, 1] and all(np.isfinite(list(l.values())).all() for _, l in seen)
    assert os.listdir(root + '/ckpt') == ['model_1']
    cfg2, tr2 = trainer()
    cfg2.training_steps = 2
    seen2 = []
    assert tr2.train(logger=lambda i, log: seen2.append(i)) == 2
    assert seen2 == [2]                                   # resumed after the checkpointed iteration
    for (k, a), b in zip(tr.model.state_dict().items(), torch.load(root + '/ckpt/model_1', weights_only=False)['model'].values()):
        assert torch.equal(a, b), k


@pytest.mark.gpu
def test_graph_replayed_training_from_feature_files_on_the_gpu(tmp_path):
    """the whole chain on the device: feature files -> MelDataset -> DeviceLoader (static shapes) -> VQGANTrainer with
    hipGraph replay across warm-up and GAN phase; losses finite, parameters move, graphs were replayed"""
    import _parity
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.utils.config import ConfigItem
    root = str(tmp_path)
    os.makedirs(root + '/mel')
    os.makedirs(root + '/wav')
    rng = np.random.default_rng(0)
    ids = []
    for i, T in enumerate((30, 41, 26, 35, 50, 24)):
        ids.append('u%d' % i)
        np.save(root + '/mel/u%d.npy' % i, rng.standard_normal((T, 80)).astype(np.float32))
        np.save(root + '/wav/u%d.npy' % i, rng.uniform(-1, 1, (T * 300, 1)).astype(np.float32))
    with open(root + '/id.list', 'w') as f:
        f.write('\n'.join(ids) + '\n')
    cfg, task = _parity.build_small('cuda:0')
    cfg.dataset = ConfigItem(dict(_name='MelDataset', id_list=root + '/id.list', feature=['mel', 'wav'], samplerate=24000,
                                  dimension=[80, 1], frameshift=[300, 1],
                                  feature_path=[root + '/mel/{}.npy', root + '/wav/{}.npy'], padding_value=[-4.0, 0.0],
                                  segment_length=24 * 300, pre_load=True))
    cfg.dataloader = ConfigItem(dict(batch_size=3, num_workers=0))
    cfg.save_checkpoint_dir, cfg.iters_per_checkpoint, cfg.training_steps = root + '/ckpt', 100, 12
    cfg.trainer.warmup_steps = 3
    tr = build_trainer(cfg, task, num_gpus=1, rank=0)
    tr.use_graphs = True
    before = {k: v.detach().clone() for k, v in tr.model.state_dict().items()}
    seen = []
    assert tr.train(logger=lambda i, log: seen.append({k: float(v) for k, v in log['loss'].items()})) == 12
    assert len(seen) == 13 and all(np.isfinite(list(l.values())).all() for l in seen)
    assert 'g_loss' in seen[-1] and 'g_loss' not in seen[0]
    assert tr._graphs is not None
    moved = sum(int(not torch.equal(v, before[k])) for k, v in tr.model.state_dict().items())
    assert moved > 100


def test_infer_entry_point_writes_waveforms_from_a_checkpoint(tmp_path):
    """infer.py: checkpoint -> task (its stored configuration) -> evaluation dataset -> analysis-synthesis -> one 16-bit
    PCM file per utterance with hop * frames samples, and the raw-float copy requested beside it"""
    import subprocess
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'tests', 'emu')])
    from msmctts_amd.hip import lib
    lib.use_library_for_tests(os.path.join(ROOT, 'tests', 'emu', 'libmsmc_emu.so'))
    import importlib.util
    import _parity
    root = str(tmp_path)
    os.makedirs(root + '/mel')
    rng = np.random.default_rng(3)
    for uid, T in (('a', 12), ('b', 20)):
        np.save(root + '/mel/%s.npy' % uid, rng.standard_normal((T, 80)).astype(np.float32))
    with open(root + '/id.list', 'w') as f:
        f.write('a\nb\n')
    cfg, task = _parity.build_small('cpu')
    conf = cfg.to_dict()
    conf['dataset'] = dict(_name='MelDataset', id_list=root + '/id.list', feature=['mel'], samplerate=24000, dimension=[80],
                           frameshift=[300], feature_path=[root + '/mel/{}.npy'], padding_value=[-4.0])
    conf['save_features'] = [['wav', '.wav', 24000], ['wav', '.dat', 24000]]
    torch.save({'model': task.state_dict(), 'iteration': 7, 'config': conf}, root + '/model_7')
    spec = importlib.util.spec_from_file_location('infer_entry', os.path.join(ROOT, 'infer.py'))
    entry = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(entry)
    argv, sys.argv = sys.argv, ['infer.py', '-m', root + '/model_7', '-j', '2']
    try:
        entry.main()
    finally:
        sys.argv = argv
    out = root + '/eval-7/wav'
    assert sorted(os.listdir(out)) == ['a.dat', 'a.wav', 'b.dat', 'b.wav']
    with wave.open(out + '/b.wav', 'rb') as w:
        assert (w.getnframes(), w.getframerate(), w.getsampwidth()) == (20 * 300, 24000, 2)
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype='<i2')
    raw = np.fromfile(out + '/b.dat', dtype=np.float32)
    assert raw.shape == (6000,) and np.abs(pcm / 32767.0 - np.clip(raw, -1, 1)).max() < 1e-4
