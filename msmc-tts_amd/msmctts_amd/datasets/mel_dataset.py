"""Acoustic-feature dataset of autoencoder training (drop-in for reference msmctts/datasets/mel_dataset.py:9-66): the
batch contract of ``VQGANTrainer.train_step`` -- ``mel (B, T, 80)`` padded with ``padding_value['mel']``, ``wav (B, T*hop, 1)``,
utterances sorted by decreasing mel length, ``mel_length``, ``wav_length = mel_length * frameshift['mel']``."""
import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence

from .base_dataset import BaseDataset, align_features


class MelDataset(BaseDataset):
    def parse_case(self, index):
        items = super().parse_case(index)
        items.update(align_features({k: v for k, v in items.items() if self.frameshift.get(k, 0) > 0}, self.frameshift))
        return items

    def collate_fn(self, batch):
        cols = {name: [torch.from_numpy(item[name]) if isinstance(item[name], np.ndarray) else item[name] for item in batch]
                for name in batch[0].keys()}
        lengths, order = torch.sort(torch.LongTensor([m.shape[0] for m in cols['mel']]), dim=0, descending=True)
        out = {}
        for name, values in cols.items():
            values = [values[i] for i in order]
            if name in ('dur', 'npw'):
                out[name + '_length'] = torch.tensor([v.shape[0] for v in values], dtype=torch.int32)
                values = [v.squeeze(-1) if v.dim() == 2 else v for v in values]
            if isinstance(values[0], torch.Tensor):
                values = (pad_sequence(values, batch_first=True, padding_value=self.padding_value[name])
                          if values[0].dim() >= 1 else torch.stack(values))
            out[name] = values
        out['mel_length'] = lengths
        if 'wav' in out:
            out['wav_length'] = lengths * self.frameshift['mel']
        return out
