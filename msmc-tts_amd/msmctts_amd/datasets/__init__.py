"""Data path (drop-in for reference msmctts/datasets/__init__.py:8-34) + the device-side half the reference leaves to
``to_model``: ``DeviceLoader`` keeps one batch ahead on the GPU (pinned staging, copy on a side HIP stream overlapping the
running step) and can pad every batch to a fixed frame count, which is what makes a step hipGraph-replayable with real
data (static shapes); it also hands the host copy of ``mel_length`` along so that the trainer's window sampling does not
read lengths back from the device."""
from os.path import dirname

import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from ..utils.utils import module_search


def build_dataset(config):
    cls = module_search(config['_name'], dirname(__file__), 'msmctts_amd.datasets')
    return cls(**{k: v for k, v in config.items() if k[:1] != '_'})


def build_dataloader(config_dataset, config_dataloader, distributed=False):
    """-> (dataset, sampler, loader): shuffled (per-rank sharded when distributed), incomplete last batch dropped"""
    dataset = build_dataset(config_dataset)
    sampler = DistributedSampler(dataset) if distributed else None
    workers = config_dataloader.num_workers
    loader = DataLoader(dataset, num_workers=workers, collate_fn=getattr(dataset, 'collate_fn', None),
                        shuffle=sampler is None, sampler=sampler, batch_size=config_dataloader.batch_size,
                        pin_memory=torch.cuda.is_available(), drop_last=True,
                        prefetch_factor=2 if workers > 0 else None, persistent_workers=False)
    return dataset, sampler, loader


class DeviceLoader(object):
    """Iterates a loader one batch ahead on ``device``.

    ``pad_frames`` (with ``hop``): ``mel`` is padded to ``[B, pad_frames, C]`` with ``mel_pad`` and ``wav`` to
    ``[B, pad_frames * hop, 1]`` with zeros whatever the longest utterance of the batch -- static shapes."""

    def __init__(self, loader, device, pad_frames=None, hop=None, mel_pad=0.0):
        self.loader, self.device = loader, torch.device(device)
        self.pad_frames, self.hop, self.mel_pad = pad_frames, hop, mel_pad
        self.stream = torch.cuda.Stream(self.device) if self.device.type == 'cuda' else None

    def __len__(self):
        return len(self.loader)

    def _static(self, batch):
        T = self.pad_frames
        if T is None or 'mel' not in batch:
            return batch
        mel = batch['mel']
        assert mel.shape[1] <= T, 'utterance of %d frames in a loader padded to %d' % (mel.shape[1], T)
        if mel.shape[1] < T:
            batch['mel'] = torch.nn.functional.pad(mel, (0, 0, 0, T - mel.shape[1]), value=self.mel_pad)
        if 'wav' in batch and self.hop:
            wav = batch['wav']
            if wav.shape[1] < T * self.hop:
                batch['wav'] = torch.nn.functional.pad(wav, (0, 0, 0, T * self.hop - wav.shape[1]))
        return batch

    def _upload(self, batch):
        batch = self._static(dict(batch))
        host_lengths = batch['mel_length'].tolist() if 'mel_length' in batch else None
        if self.stream is None:
            out = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        else:
            with torch.cuda.stream(self.stream):
                out = {k: (v.pin_memory().to(self.device, non_blocking=True) if torch.is_tensor(v) and not v.is_pinned()
                           else v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
        if host_lengths is not None:
            out['mel_length_host'] = host_lengths
        return out

    def __iter__(self):
        ahead = None
        for batch in self.loader:
            nxt = self._upload(batch)
            if ahead is not None:
                yield self._ready(ahead)
            ahead = nxt
        if ahead is not None:
            yield self._ready(ahead)

    def _ready(self, batch):
        if self.stream is not None:                     # the consumer's stream waits for the copy, not the host
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.stream)
            for v in batch.values():
                if torch.is_tensor(v):
                    v.record_stream(cur)
        return batch
