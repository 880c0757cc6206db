"""``task._name`` classes of the shipped YAMLs (reference msmctts/tasks/msmc_tts.py:10-151).

``MSMCTTS`` is the container whose children (``autoencoder``, ``discriminator``, ``predictor``) key optimizers and
checkpoints in training, and the analysis-synthesis / text-to-waveform glue of ``infer.py``:
``task._mode: train_autoencoder`` -> mel -> MSMC-VQ-GAN -> waveform, ``train_predictor`` -> text -> multi-stage predictor
-> quantised stage features -> the (frozen, separately checkpointed) autoencoder's ``synthesis`` -> waveform.  The
acoustic-model + external-vocoder pipeline of the reference's ``TTS.infer_step`` (:21-84) belongs to other YAMLs and is not
part of this package.
"""
import torch

from .base_task import BaseTask


class MSMCTTS(BaseTask):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        ds = self.config.dataset if 'dataset' in self.config else None
        self.samplerate = ds.samplerate if ds is not None else None
        self.fs = {f: s for f, s in zip(ds.feature, ds.frameshift)} if ds is not None else {}
        self.training_mode = self.config.task._mode if '_mode' in self.config.task else 'train_autoencoder'
        self.load_modules = False

    def train_step(self, input_dict, mode=None):
        mode = self.training_mode if mode is None else mode
        if mode == 'train_autoencoder':
            return self.analysis_synthesis(input_dict)
        if mode == 'train_predictor':
            return self.predict(input_dict)
        raise ValueError('unknown task mode %r' % (mode,))

    def infer_step(self, input_dict, mode=None):
        mode = self.training_mode if mode is None else mode
        if mode == 'train_predictor' and not self.load_modules:
            self.pre_infer()
        return self.train_step(input_dict, mode)

    def analysis_synthesis(self, input_dict):
        """mel (B, T, 80) + mel_length -> {'wav': (B, T * hop)}"""
        out = self.autoencoder(**{k: v for k, v in input_dict.items() if k in ('mel', 'mel_length')})
        return {'wav': out['decoder_outputs'].squeeze(-1)}

    def predict(self, input_dict):
        """text (+ durations) -> per-stage features -> waveform; utterances trimmed to their predicted length"""
        feed = {k: v for k, v in input_dict.items() if k not in ('mel', 'mel_length')}
        out = self.predictor(**feed)
        feats, lengths = out['feat'], out['feat_length']
        wavs = self.autoencoder.synthesis(feats, lengths)[..., 0]
        wav_lengths = (lengths[-1] * wavs.shape[1] / feats[-1].shape[1]).int()
        out['wav'] = [w[:n] for w, n in zip(wavs, wav_lengths)]
        out['embedding'] = feats[-1]
        return out

    def pre_infer(self):
        """the frozen analyser / synthesiser named by ``task.autoencoder._checkpoint`` (and ``_config``); the predictor gets
        its quantisers so that predictions are snapped to codewords"""
        from . import load_model
        self.load_modules = True
        acfg = self.config.task.autoencoder if 'autoencoder' in self.config.task else None
        if acfg is not None and '_checkpoint' in acfg:
            model = load_model('autoencoder', acfg._checkpoint, acfg._config if '_config' in acfg else None)
            self.autoencoder = model.to(next(self.parameters()).device).eval()
        if hasattr(self, 'predictor') and hasattr(self, 'autoencoder'):
            self.predictor.autoencoder = self.autoencoder


class TTS(MSMCTTS):
    """``task._name: TTS`` resolves to the same container (its acoustic-model + vocoder pipeline is out of scope)."""
