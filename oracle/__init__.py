"""CPU oracle for the MSMC-VQ-GAN training hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch / numpy / C restatement of the reference
algorithm (hhguo/MSMC-TTS @ v2) for the path BASELINE.json's north_star names:
one ``VQGANTrainer.train_step`` (reference msmctts/trainers/msmctts_trainer.py:115-209)
over MSMCVQGAN + UnivNetDiscriminator.  Every function cites the reference
file:line it follows.  Nothing here is imported by the product package
``msmc-tts_amd/msmctts_amd``: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.

Parity status
-------------
The reference ships no tests, golden vectors or checkpoints for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, imported in the build container through the import shims in
``tests/golden/_ref_shims.py`` and frozen as fixtures under ``tests/golden/*.npz`` by
``tests/golden/make_golden.py`` (committed; the reference never travels).
``tests/test_oracle_vs_golden.py`` checks every fixture on CPU.

One third-party piece is restated rather than imported: ``librosa.filters.mel``
(requirements.txt pins ``librosa>=0.8.0``; call site criterions/stft_loss.py:85) is
not installed in the image.  ``oracle.audio.slaney_mel_basis`` follows librosa's
published algorithm (Slaney scale, Slaney area normalisation); for that one
matrix parity is "unpinned" against librosa itself and pinned only against
librosa's documented definition.
"""
