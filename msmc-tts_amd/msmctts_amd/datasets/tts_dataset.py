"""Text + acoustic dataset of predictor training (drop-in for reference msmctts/datasets/tts_dataset.py:9-99): batch sorted
by decreasing text length; durations given in seconds are converted to frames with the rounding error carried forward, and
their sum is reconciled with the mel length (at most 5 frames apart)."""
import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence

from .base_dataset import BaseDataset, align_features


class TTSDataset(BaseDataset):
    def parse_case(self, index):
        items = super().parse_case(index)
        items.update(align_features({k: v for k, v in items.items() if self.frameshift.get(k, 0) > 0}, self.frameshift))
        text = items['text']
        if text.ndim == 2 and text.shape[1] == 1:
            text = items['text'] = text[:, 0]
        if 'dur' in items:
            durs = items['dur'].squeeze(1) if items['dur'].ndim == 2 else items['dur']
            assert len(durs) == len(text), '%s : %d v.s. %d' % (self.id_list[index], len(durs), len(text))
            if 'mel' in items:
                n = items['mel'].shape[0]
                if n / sum(durs) > 100:                      # seconds -> frames, carrying the rounding error forward
                    durs = durs * self.samplerate / self.frameshift['mel']
                    for i in range(len(durs)):
                        whole = round(durs[i])
                        if i < len(durs) - 1:
                            durs[i + 1] += durs[i] - whole
                        durs[i] = whole
                slack = n - sum(durs)
                assert -5 <= slack <= 5, '%s: %d v.s. %s' % (self.id_list[index], n, sum(durs))
                durs[-1] += slack
            items['dur'] = durs
        return items

    def collate_fn(self, batch):
        cols = {name: [torch.from_numpy(item[name]) if isinstance(item[name], np.ndarray) else item[name] for item in batch]
                for name in batch[0].keys()}
        lengths, order = torch.sort(torch.LongTensor([t.shape[0] for t in cols['text']]), dim=0, descending=True)
        cols = {name: [values[i] for i in order] for name, values in cols.items()}
        if 'speaker' in cols:
            cols['speaker'] = torch.Tensor(cols['speaker'])
        cols['text_length'] = lengths
        for name in ('text', 'tone', 'dur'):
            if name in cols:
                cols[name] = pad_sequence(cols[name], batch_first=True, padding_value=self.padding_value[name])
        for name in ('mel', 'wav', 'pitch', 'energy'):
            if name not in cols:
                continue
            if name in ('mel', 'wav'):
                cols[name + '_length'] = torch.Tensor([v.shape[0] for v in cols[name]])
            cols[name] = pad_sequence(cols[name], batch_first=True, padding_value=self.padding_value[name])
        return cols
