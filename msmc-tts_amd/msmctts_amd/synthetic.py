"""Seeded synthetic batches with the contract of ``MelDataset.collate_fn`` + ``to_model``
(reference msmctts/datasets/mel_dataset.py:25-56, utils/utils.py:137-151).

``mel (B,T,in_dim)`` ~ N(0,1) with padding value -4, ``mel_length (B,) int64`` sorted descending with
``max == T``, ``wav (B, T*hop, 1)`` ~ U(-1,1) with padding 0, ``wav_length = mel_length * hop``.
"""
import torch


def make_batch(batch_size=16, frames=400, in_dim=80, hop=300, seed=1234, rank=0, device='cpu'):
    g = torch.Generator().manual_seed(seed + rank)
    lengths = torch.randint(frames // 2, frames + 1, (batch_size,), generator=g)
    lengths[0] = frames
    lengths = torch.sort(lengths, descending=True).values.to(torch.int64)
    mel = torch.randn(batch_size, frames, in_dim, generator=g)
    wav = torch.rand(batch_size, frames * hop, 1, generator=g) * 2 - 1
    t = torch.arange(frames)[None, :, None]
    mel = torch.where(t < lengths[:, None, None], mel, torch.full_like(mel, -4.0))
    s = torch.arange(frames * hop)[None, :, None]
    wav = torch.where(s < (lengths * hop)[:, None, None], wav, torch.zeros_like(wav))
    batch = {'mel': mel, 'mel_length': lengths, 'wav': wav, 'wav_length': lengths * hop}
    return {k: v.to(device) for k, v in batch.items()}


def make_text_batch(mel_length, n_symbols=(100, 10, 2), phonemes=(30, 56), seed=4321, rank=0, device='cpu'):
    """The text side of a ``TTSDataset`` batch (reference msmctts/datasets/tts_dataset.py) for the given mel lengths:
    ``text (B, P, 3)`` symbol / tone / boundary ids (0 = padding), ``text_length (B,)``, ``dur (B, P)`` whole-frame durations of
    at least one frame that sum to the utterance's ``mel_length`` -- what the predictor's length regulator expands by."""
    g = torch.Generator().manual_seed(seed + rank)
    lengths = [int(n) for n in mel_length]
    B = len(lengths)
    text_length = torch.randint(phonemes[0], phonemes[1] + 1, (B,), generator=g)
    text_length = torch.minimum(text_length, torch.tensor(lengths)).to(torch.int64)
    P = int(text_length.max())
    text = torch.zeros(B, P, len(n_symbols), dtype=torch.int64)
    dur = torch.zeros(B, P, dtype=torch.int64)
    for b, (tl, ml) in enumerate(zip(text_length.tolist(), lengths)):
        for i, n in enumerate(n_symbols):
            text[b, :tl, i] = torch.randint(1, n, (tl,), generator=g)
        cuts = sorted(torch.randperm(ml - 1, generator=g)[:tl - 1].add(1).tolist())
        edges = [0] + cuts + [ml]
        dur[b, :tl] = torch.tensor([edges[i + 1] - edges[i] for i in range(tl)])
    batch = {'text': text, 'text_length': text_length, 'dur': dur}
    return {k: v.to(device) for k, v in batch.items()}
