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
