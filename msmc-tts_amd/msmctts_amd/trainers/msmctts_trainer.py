"""VQ-GAN training step (re-expression of reference msmctts/trainers/msmctts_trainer.py:39-219).

Same phase logic, loss weights, loss-dictionary keys and optimizer order as the reference:
warm-up (no vocoder) for ``iteration < warmup_steps``; at ``iteration == warmup_steps`` the vocoder
runs but no GAN/STFT loss is taken (strict ``>``, :146); afterwards mel/STFT loss, D step (fake
detached + real, LSGAN), then the G step against the *updated* D (adversarial + feature matching),
gradient clipping on the autoencoder only.

Differences that change no result:
* in the G step the reference also back-propagates into D's weights and throws those gradients
  away; here D is frozen for that pass, D(real) runs without a graph and nothing of D is
  all-reduced on the G backward;
* loss values stay on the device (0-dim tensors, ``float(v)`` to read) instead of ~15 host syncs.
"""
import contextlib
import os
import random

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..hip import convnet as hipconvnet
from ..hip import losses as hiploss
from ..hip import vq as hipvq
from ..utils.utils import get_mask_from_lengths
from .base_trainer import BaseTrainer
from .criterions.stft_loss import MelLoss, MultiResolutionSTFTLoss


_ONES = {}


def _one_like(loss):
    """the seed gradient of ``loss.backward()``: autograd otherwise allocates and fills a ones tensor per call (a launch per
    backward pass, replayed with every step); one constant per (device, dtype), created on first use -- the eager warm-up
    steps, before any capture"""
    key = (loss.device, loss.dtype)
    one = _ONES.get(key)
    if one is None:
        assert not (loss.is_cuda and torch.cuda.is_current_stream_capturing()), 'the seed gradient must exist before the capture'
        one = _ONES[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
    return one


class QuantizerLoss(nn.Module):
    def __init__(self, lambda_vq=1, lambda_pr=1):
        super().__init__()
        self.lambda_vq, self.lambda_pr = lambda_vq, lambda_pr

    def forward(self, outputs):
        loss = {}
        parts, weights = [], []                 # vq_loss = sum of weights[i] * parts[i]: ONE launch (hiploss.weighted_sum)
        diffs = outputs['encoder_diffs']
        if not isinstance(diffs, (tuple, list)):
            diffs = [diffs]
        for i, terms in enumerate(diffs):
            length = outputs['encoder_lengths'][i]
            terms = terms if isinstance(terms, (tuple, list)) else [terms]
            for j, term in enumerate(terms):
                if _fused_masked(term, length):
                    term = hiploss.masked_mean(term, length)       # sum over valid rows / (sum of lengths * channels), two launches
                else:
                    pad = get_mask_from_lengths(length.to(term.device), term.shape[1]).unsqueeze(-1)
                    term = term.masked_fill(pad, 0).sum() / length.sum() / term.shape[2]
                loss['latent_loss_{}_{}'.format(i, j)] = term
                parts.append(term)
                weights.append(self.lambda_vq)
        dd = outputs.get('decoder_diffs')
        if isinstance(dd, dict):
            dd = dict(dd)
            parts.append(dd.pop('total_loss'))
            weights.append(self.lambda_pr)
            loss.update(dd)
        loss['vq_loss'] = hiploss.weighted_sum(parts, weights) if parts else 0
        return loss


def _fused_masked(x, lengths):
    """the length-masked mean runs as csrc/losses.hip's masked_mean (GPU, or the interpreter in the CPU tests)"""
    return (torch.is_tensor(x) and x.dim() == 3 and x.dtype in (torch.float32, torch.bfloat16) and hiploss.usable(x)
            and torch.is_tensor(lengths) and lengths.device == x.device and lengths.dtype in (torch.int32, torch.int64))


@contextlib.contextmanager
def _frozen(module):
    flags = [(p, p.requires_grad) for p in module.parameters()]
    for p, _ in flags:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p, f in flags:
            p.requires_grad_(f)


class VQGANTrainer(BaseTrainer):
    def __init__(self, config, model, num_gpus=1, rank=0, warmup_steps=0, lambda_frame=1.0,
                 eval_inteval_iters=1000, grad_clip_thresh=1.0, sample_lengths=24000, lambda_vq=1, lambda_pr=1,
                 lambda_fm=2, lambda_stft=45, stft_loss_func='mel_loss', stft_loss_config=None,
                 sync_codebook_stats=False):
        super().__init__(config, model, num_gpus, rank)
        # Not in the reference (its ranks EMA-update their VQ codebooks from the local batch and rank 0's copy is the
        # one checkpointed): sum the EMA statistics over ranks -- one small all-reduce per step -- so that every rank
        # holds the codebooks of the global batch.
        self.sync_codebook_stats = bool(sync_codebook_stats)
        for m in self.model.modules():
            if hasattr(m, 'decay') and hasattr(m, 'n_embed'):
                m.sync_stats = self.sync_codebook_stats
        self.lambda_frame, self.warmup_steps = lambda_frame, warmup_steps
        self.frameshift = self.config.dataset.frameshift[self.config.dataset.feature.index('mel')]
        self.frame_lengths = -1 if sample_lengths == -1 else sample_lengths // self.frameshift
        self.eval_inteval_iters, self.grad_clip_thresh = eval_inteval_iters, grad_clip_thresh
        self.vq_criterion = QuantizerLoss(lambda_vq=lambda_vq, lambda_pr=lambda_pr)
        self.sample_lengths, self.lambda_fm, self.lambda_stft = sample_lengths, lambda_fm, lambda_stft
        if stft_loss_func == 'mel_loss':
            sr = config.dataset.samplerate
            kw = dict(sample_rate=sr, win_size=sr // 20, hop_size=sr // 80, num_mels=128)
            kw['fft_size'] = 2048 if kw['win_size'] > 1024 else 1024
            if stft_loss_config is not None:
                kw.update(stft_loss_config)
            self.stft_criterion = MelLoss(**kw)
        elif stft_loss_func == 'mr_stft':
            self.stft_criterion = MultiResolutionSTFTLoss(**dict(stft_loss_config or {}))
        self.rng = random              # python global RNG, like the reference (:214); tests inject their own
        self._amp_applied = None
        # the generator step reads the spectral front-end images the D step built from the same waveforms (A/B: 0)
        self.reuse_fronts = os.environ.get('MSMC_REUSE_FRONTS', '1') != '0'
        # more parallel branches of the step (streams of the library's own, hip/convnet.py own_streams): the spectral loss of
        # the prediction next to the discriminator step, the no-gradient D(real) pass of the generator step next to D(fake)
        self.loss_fork = os.environ.get('MSMC_LOSS_FORK', '1') != '0'
        self.real_fork = os.environ.get('MSMC_REAL_FORK', '1') != '0'
        # hipGraph mode under data parallelism: 'serial' = one flat all-reduce per child between the replayed segments;
        # 'overlap' = the reducer's bucketed all-reduces captured INTO the segments (distributed/distributed.py docstring)
        self.graph_exchange = os.environ.get('MSMC_GRAPH_EXCHANGE', 'serial')
        self.use_graphs = False        # replay the step from hipGraphs (static shapes): the GAN phase as three segments, the
        self._graphs = None            # warm-up phase (the reference trains 50 000 steps in it, msmc_vq_gan.yaml:91) as two
        self._graphs_warm = None
        self.graph_warmup = os.environ.get('MSMC_GRAPH_WARMUP', '1') != '0'     # A/B: 0 keeps the warm-up phase eager
        self.amp_autocast = True       # False: only the HIP conv stacks compute in amp_dtype, stock operators stay fp32
        self.amp_dtype = None          # e.g. torch.bfloat16: autocast for the GEMM/conv bodies (VQ search stays fp32)

    def random_select(self, mel_length):
        lengths = mel_length.tolist() if torch.is_tensor(mel_length) else list(mel_length)
        frame_windows, sample_windows = [], []
        for n in lengths:
            start = self.rng.randrange(max(1, int(n) - self.frame_lengths))
            end = start + self.frame_lengths
            frame_windows.append((start, end))
            sample_windows.append((start * self.frameshift, end * self.frameshift))
        return frame_windows, sample_windows

    def _amp(self):
        if self._amp_applied is not self.amp_dtype:          # tell the HIP conv stacks their compute dtype
            for m in self.model.modules():
                if hasattr(m, 'hip_dtype'):
                    m.hip_dtype = self.amp_dtype or torch.float32
            if hasattr(self, 'stft_criterion'):      # (bf16 runs: its constant-matrix GEMMs as split-bf16 products)
                self.stft_criterion.hip_dtype = self.amp_dtype or torch.float32
            self._amp_applied = self.amp_dtype
        if self.amp_dtype is None or not self.amp_autocast:
            return contextlib.nullcontext()
        device_type = next(self.model.parameters()).device.type
        return torch.autocast(device_type=device_type, dtype=self.amp_dtype)

    # ------------------------------------------------------------------------------------------
    # The step is three segments separated by the two gradient exchanges of the data-parallel path:
    #   A  autoencoder forward, losses, D(fake.detach) / D(real), d_loss.backward()
    #   B  D optimizer step, D(fake) / D(real) against the updated D, g_loss.backward()
    #   C  gradient clipping, autoencoder optimizer step
    # Eagerly they run back to back; ``use_graphs`` captures each into a hipGraph (static batch buffers,
    # window indices on the device) and replays them with the RCCL all-reduces in between.
    # ------------------------------------------------------------------------------------------
    def _segment_a(self, st):
        losses = st.losses = {}
        mel, mel_length = st.mel, st.mel_length
        ae, disc = self.model.autoencoder, getattr(self.model, 'discriminator', None)
        with self._amp():
            out = ae(mel, mel_length, warmup=st.phase == 0, window=st.frame_window)
        vq = self.vq_criterion(out)
        losses.update(vq)
        g_loss = vq['vq_loss']
        if 'mel_outputs' in out:
            if _fused_masked(out['mel_outputs'], mel_length) and mel.dtype == torch.float32 and mel.is_contiguous():
                # masked mean of (mel - mel_outputs)^2, the prediction read in its own dtype (no cast pass)
                ml = hiploss.masked_mean(out['mel_outputs'], mel_length, b=mel)
            else:
                ml = F.mse_loss(mel, out['mel_outputs'].float(), reduction='none')
                ml = ml.masked_fill(get_mask_from_lengths(mel_length, ml.shape[1]).unsqueeze(-1), 0)
                ml = ml.sum() / mel_length.sum() / ml.shape[2]
            losses['frame_loss'] = ml
            g_terms, g_weights = [g_loss, ml], [1.0, self.lambda_frame]
        else:
            g_terms, g_weights = [g_loss], [1.0]
        if st.phase < 2:
            st.g_loss = hiploss.weighted_sum(g_terms, g_weights) if len(g_terms) > 1 else g_loss
            return
        st.predict = predict = out['decoder_outputs'].squeeze(-1).float()
        target = st.target

        def spectral_loss():
            stl = self.stft_criterion(predict, target)
            if isinstance(stl, dict):
                for name, term in stl.items():
                    losses[name] = term
                stl = sum(stl.values())
            return stl
        # the spectral loss has no consumer before the generator step: a side branch under the discriminator step (its
        # backward nodes replay on the same stream, next to the generator step's D(fake) backward)
        side = (hipconvnet.own_streams(predict.device, 1, 'loss-fork') if (self.loss_fork and predict.is_cuda
                                                                             and hipconvnet.STREAMS_ENABLED) else [])
        if side:
            main = torch.cuda.current_stream(predict.device)
            side[0].wait_stream(main)
            with torch.cuda.stream(side[0]):
                stl = spectral_loss()
        else:
            stl = spectral_loss()
        # D(fake.detach()) and D(real) as ONE pass over the concatenated batch (every layer is per-sample, so
        # the scores are those of two separate passes): half the launches, twice the work per launch
        B = predict.shape[0]
        both = torch.cat((predict.detach(), target), dim=0)
        # the resolution discriminators' framed-DFT images of [fake; real], kept for the generator step (same waveforms,
        # no parameters in the front-end): computed here once per step
        st.fronts = disc.spectral_fronts(both) if (self.reuse_fronts and hasattr(disc, 'spectral_fronts')) else None
        with self._amp():
            both_scores, _ = disc(both, fronts=st.fronts) if st.fronts is not None else disc(both)
        # LSGAN terms of the two halves, summed over the 10 sub-discriminators, straight from the [2B] score tensors
        d_fake, d_real = hiploss.mse_const_halves(both_scores, B, 0.0, 1.0)
        d_loss = hiploss.weighted_sum([d_real, d_fake])
        losses['d_loss_real'], losses['d_loss_fake'], losses['d_loss'] = d_real, d_fake, d_loss
        self.optimizer.zero_grad(['discriminator'])
        d_loss.backward(gradient=_one_like(d_loss))
        if side:
            main.wait_stream(side[0])
        losses['stft_loss'] = stl
        st.g_loss = hiploss.weighted_sum(g_terms + [stl], g_weights + [self.lambda_stft])     # vq + frame + stft in one launch

    def _segment_b(self, st):
        losses, disc = st.losses, getattr(self.model, 'discriminator', None)
        if st.phase == 2:
            self.optimizer.step(['discriminator'])
            # generator step against the updated D; D's own gradients are not needed
            if getattr(st, 'fronts', None) is not None:
                B = st.predict.shape[0]
                def real_pass():
                    with torch.no_grad():
                        return disc(st.target, fronts=st.fronts.rows(B, 2 * B))[1]
                side = (hipconvnet.own_streams(st.predict.device, 1, 'real-pass')
                        if (self.real_fork and st.predict.is_cuda) else [])
                if side and hasattr(disc, 'prepare_weights'):
                    disc.prepare_weights()          # (the optimizer just stepped: refresh the weight images BEFORE the fork)
                with _frozen(disc), self._amp():
                    real_feats, (fake_scores, fake_feats) = hipconvnet.fork_join(
                        side, [real_pass], inputs=(st.target,),
                        main_thunk=lambda: disc(st.predict, fronts=st.fronts.rows(0, B, wav=st.predict)))
                st.fronts = None
            else:
                with _frozen(disc), self._amp():
                    fake_scores, fake_feats = disc(st.predict)
                    with torch.no_grad():
                        _, real_feats = disc(st.target)
            adv = hiploss.mse_const_sum(fake_scores, 1.0)
            fm = hiploss.l1_sum([a for fa in fake_feats for a in fa], [b for fb in real_feats for b in fb])
            if self.lambda_fm != 'auto':
                adv = hiploss.weighted_sum([adv, fm], [1.0, self.lambda_fm])
            else:
                adv = adv + fm * (st.g_loss / fm).detach()
            st.g_loss = hiploss.weighted_sum([st.g_loss, adv])
            losses['fm_loss'], losses['adv_loss'], losses['g_loss'] = fm, adv, st.g_loss
        self.optimizer.zero_grad(['autoencoder'])
        st.g_loss.backward(gradient=_one_like(st.g_loss))
        if st.g_loss.is_cuda and st.phase == 2 and hipconvnet.STREAMS_ENABLED:
            # (the side branches' backward nodes hand their results to nodes of this stream, which orders them already; the
            # explicit join costs nothing and does not depend on that)
            main = torch.cuda.current_stream(st.g_loss.device)
            for role, on in (('loss-fork', self.loss_fork), ('real-pass', self.real_fork)):
                if on:
                    main.wait_stream(hipconvnet.own_streams(st.g_loss.device, 1, role)[0])

    def _segment_c(self, st):
        if hasattr(self.optimizer, 'clip_and_step'):           # clip + AdamW of all tensors in three launches
            self.grad_norm = self.optimizer.clip_and_step('autoencoder', self.grad_clip_thresh)
            return
        self.grad_norm = nn.utils.clip_grad_norm_(self.model.autoencoder.parameters(), self.grad_clip_thresh)
        self.optimizer.step(['autoencoder'])

    def _phase(self, iteration):
        """0 = warm-up (no vocoder), 1 = vocoder runs but no GAN/STFT loss (iteration == warmup_steps), 2 = GAN."""
        if iteration < self.warmup_steps:
            return 0
        return 2 if iteration > self.warmup_steps else 1

    def _graphed_phase(self, phase):
        return bool(self.use_graphs) and (phase == 2 or (phase == 0 and self.graph_warmup))

    def replays(self, iteration):
        return self._graphed_phase(self._phase(iteration))

    def train_step(self, batch, iteration):
        phase = self._phase(iteration)
        if self._graphed_phase(phase):
            return self._train_step_graphed(batch, phase)
        reducer = getattr(self.model, 'grad_reducer', None)
        if reducer is not None:
            reducer.hooks_enabled = True         # eager step: bucketed all-reduce from the gradient hooks
        st = _StepState()
        st.phase, st.mel, st.mel_length = phase, batch['mel'], batch['mel_length']
        st.frame_window = st.target = None
        if phase > 0:
            frame_windows, sample_windows = self.random_select(batch.get('mel_length_host', batch['mel_length']))
            st.frame_window = frame_windows
            st.target = torch.stack([batch['wav'][i, s:e] for i, (s, e) in enumerate(sample_windows)], dim=0).squeeze(-1)
        self._segment_a(st)
        self._sync_codebooks()
        if phase == 2:
            self._sync_grads()
        self._segment_b(st)
        self._sync_grads()
        self._segment_c(st)
        return {'loss': {k: (v.detach() if torch.is_tensor(v) else v) for k, v in st.losses.items()}}

    # -- hipGraph replay of the GAN-phase step -----------------------------------------------------
    def _train_step_graphed(self, batch, phase=2):
        g = self._graphs if phase == 2 else self._graphs_warm
        reducer = getattr(self.model, 'grad_reducer', None)
        if reducer is not None:
            reducer.hooks_enabled = False        # (serial exchange: no collectives inside capture; overlap: _capture arms them)
        # the captured window gather reads wav[start * frameshift : (start + frame_lengths) * frameshift] unchecked: a waveform
        # shorter than its mel says (wrong hop in the data) must fail here, on the host, not as a GPU memory fault
        if phase == 2:
            lengths = batch.get('mel_length_host')
            if lengths is None:                  # (a device read-back: loaders hand the host copy along, datasets/DeviceLoader)
                lengths = batch['mel_length'].tolist()
            have, need = batch['wav'].numel() // len(lengths), (int(max(lengths)) - 1) * self.frameshift      # (windows end at frame n - 1)
            if have < need:
                raise ValueError('batch["wav"] holds %d samples per utterance, mel_length x frameshift (%d) asks for %d'
                                 % (have, self.frameshift, need))
        if g is None:
            if phase == 2 and self._graphs_warm is not None:
                # training has left the warm-up phase for good (iterations only grow): its graphs, static batch, gradient
                # tensors, memory pool and optimizer tables go BEFORE the GAN phase allocates its own -- the peak at step
                # warmup_steps + 1 is one phase's working set, not two (an out-of-memory there would come 50k steps into a run)
                self._graphs_warm = None
                if hasattr(self.optimizer, 'release'):
                    self.optimizer.release(None, 'phase0')
                import gc as _gc
                _gc.collect()
            g = self._capture(batch, phase)
            if phase == 2:
                self._graphs = g
            else:
                self._graphs_warm = g
        st = g['state']
        if phase == 2:
            starts = [self.rng.randrange(max(1, int(n) - self.frame_lengths)) for n in lengths]
            g['starts_host'].copy_(torch.tensor(starts, dtype=torch.int64))
            g['starts'].copy_(g['starts_host'], non_blocking=True)
        if batch['mel'].data_ptr() != st.mel.data_ptr():
            st.mel.copy_(batch['mel'], non_blocking=True)
            st.mel_length.copy_(batch['mel_length'], non_blocking=True)
            if phase == 2:
                g['wav'].copy_(batch['wav'].reshape(g['wav'].shape), non_blocking=True)
        hipconvnet.refresh_stale_banks()               # (parameters changed behind the graphs' back: checkpoint load ...)
        g['a'].replay()
        self._sync_codebooks(g['codebooks'])
        if phase == 2 and not g['overlap']:
            self._sync_grads_static('discriminator', g)
        g['b'].replay()
        if not g['overlap']:
            self._sync_grads_static('autoencoder', g)
        g['c'].replay()
        hipconvnet.graphs_replayed()                 # (an eager pass after this must refresh its weight images)
        vec = g['loss_vec'].clone()                  # the graph's outputs are static buffers: hand out a snapshot
        return {'loss': {k: vec[i] for i, k in enumerate(g['loss_keys'])}}

    def _sync_codebooks(self, pending=None):
        """sync_codebook_stats: cross-rank sum of the EMA statistics the forward collected, then the codebook updates
        (``pending``: the stages recorded at capture -- a replayed forward refills their static buffers)"""
        if self.sync_codebook_stats:
            hipvq.flush_codebook_sync(pending)

    def _sync_grads_static(self, child, g=None):
        """Gradient exchange between two replayed segments.  The gradients are the STATIC tensors the graphs write
        (recorded at capture): a training loop that sets ``p.grad = None`` between steps cannot hide them."""
        reducer = getattr(self.model, 'grad_reducer', None)
        if reducer is not None:
            grads = g['grads'][child] if g is not None and 'grads' in g else None
            reducer.allreduce_child(getattr(self.model, child), grads=grads)

    # -- the eager warm-up inside _capture really trains (parameters, VQ codebooks, optimizer moments move): undo it,
    #    so that a graphed run performs exactly the optimizer steps an eager run performs
    def _snapshot_state(self):
        """by NAME: modules may re-register their buffers during the first forward (the quantiser packs its per-head
        codebooks into one tensor), so tensor identities taken before the warm-up can go stale"""
        opt = []
        for o in self.optimizer.optimizers.values():
            if hasattr(o, '_ensure_state'):        # HipAdamW: the flat state must exist (and be what is recorded)
                for group in o.param_groups:       # before the warm-up, or the roll-back restores into orphans
                    o._ensure_state(group)
            for st in o.state.values():
                opt.extend(v for v in st.values() if torch.is_tensor(v))
        return {'model': {k: v.detach().clone() for k, v in self.model.state_dict().items()},
                'opt': [(t, t.detach().clone()) for t in opt]}

    def _restore_state(self, snap):
        with torch.no_grad():
            live = self.model.state_dict()                 # aliases the registered parameters / buffers
            for k, saved in snap['model'].items():
                live[k].copy_(saved)
            for t, saved in snap['opt']:
                t.copy_(saved)
            known = set(id(t) for t, _ in snap['opt'])
            for o in self.optimizer.optimizers.values():
                for st in o.state.values():
                    for v in st.values():
                        if torch.is_tensor(v) and id(v) not in known:       # moments / step created by the warm-up
                            v.zero_()

    def _build_windows(self, g, st):
        """Frame / sample index tensors from the per-utterance window starts (device arithmetic only: one launch)."""
        from ..hip import spectral as hipspectral
        starts, wav = g['starts'], g['wav']
        fl = self.frame_lengths
        if wav.dtype == torch.float32 and wav.is_contiguous() and (wav.is_cuda or hipspectral.lib._host_pointers_ok):
            st.frame_window, st.target = hipspectral.window_gather(starts, wav, fl, self.frameshift)
            return
        st.frame_window = starts.unsqueeze(1) + torch.arange(fl, device=starts.device).unsqueeze(0)
        sidx = (starts * self.frameshift).unsqueeze(1) + torch.arange(fl * self.frameshift,
                                                                      device=starts.device).unsqueeze(0)
        st.target = torch.gather(wav, 1, sidx)

    def _capture(self, batch, phase=2):
        """Warm up eagerly on a side stream, then record segments A, B, C into three graphs sharing one pool.  ``phase`` 2: the
        GAN-phase step; 0: the warm-up phase (autoencoder forward + losses | backward | clip + update -- no vocoder window, no
        discriminator; the same three segments, so the gradient exchange of the data-parallel path sits where it sits in
        phase 2: between B and C).  Each phase owns its graphs, static batch buffers, gradient tensors and memory pool."""
        dev = batch['mel'].device
        B = batch['mel'].shape[0]
        from ..hip import graphs as hipgraphs
        if not hipgraphs.memset_nodes_ordered(dev):
            raise RuntimeError(hipgraphs.HINT)
        snap = self._snapshot_state()
        st = _StepState()
        st.phase = phase
        st.mel, st.mel_length = batch['mel'].clone(), batch['mel_length'].clone()
        st.frame_window = st.target = None
        g = {'state': st}
        if phase == 2:
            g.update(wav=batch['wav'].reshape(B, -1).clone(), starts=torch.zeros(B, dtype=torch.int64, device=dev),
                     starts_host=torch.zeros(B, dtype=torch.int64).pin_memory())

        def run_eager():
            if phase == 2:
                self._build_windows(g, st)
            self._segment_a(st)
            hipvq.flush_codebook_sync(local=True)
            self._segment_b(st)                      # (no gradient exchange: the warm-up's updates are rolled back)
            self._segment_c(st)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self.model.zero_grad(set_to_none=True)
                run_eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # Drop every reference to the warm-up autograd graphs (their AccumulateGrad nodes are bound to a
        # stream) and capture on the SAME side stream the warm-up ran on, so that gradient accumulation is
        # recorded into the graphs instead of running on another stream.
        for name in ('losses', 'g_loss', 'predict', 'fronts') + (('target', 'frame_window') if phase == 2 else ()):
            setattr(st, name, None)
        self.grad_norm = None
        import gc as _gc
        _gc.collect()
        self.model.zero_grad(set_to_none=True)          # gradients get (static) graph-pool storage during capture
        ga, gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # with a process group alive, RCCL's watchdog thread polls events while we capture: only this thread's calls
        # may invalidate the capture
        import torch.distributed as dist
        mode = 'thread_local' if dist.is_available() and dist.is_initialized() else 'global'
        reducer = getattr(self.model, 'grad_reducer', None)
        overlap = g['overlap'] = reducer is not None and self.graph_exchange == 'overlap'

        def exchange_in_graph():
            """overlap mode: the bucketed all-reduces the hooks issued during this segment's backward are part of the capture
            (a branch on RCCL's stream); join them and write the averages back, still inside the capture"""
            if overlap:
                reducer.finish()
                reducer.hooks_enabled = False
        with torch.cuda.graph(ga, stream=side, capture_error_mode=mode):
            if phase == 2:
                self._build_windows(g, st)
            if overlap and phase == 2:
                reducer.hooks_enabled = True
            self._segment_a(st)
            if phase == 2:
                exchange_in_graph()
        g['codebooks'] = list(hipvq.PENDING)    # (sync_codebook_stats: the stages whose statistics segment A refills)
        del hipvq.PENDING[:]
        torch.cuda.synchronize()
        tag = 'phase%d' % phase
        prepare = getattr(self.optimizer, 'prepare', lambda names=None, tag=None: None)
        if phase == 2:
            prepare(['discriminator'], tag)     # (tensor tables over the static gradients segment A just allocated)
        with torch.cuda.graph(gb, pool=ga.pool(), stream=side, capture_error_mode=mode):
            if overlap:
                reducer.hooks_enabled = True
            self._segment_b(st)
            exchange_in_graph()
        keys = [k for k, v in st.losses.items() if torch.is_tensor(v)]
        torch.cuda.synchronize()
        prepare(['autoencoder'], tag)
        with torch.cuda.graph(gc, pool=ga.pool(), stream=side, capture_error_mode=mode):
            self._segment_c(st)
            loss_vec = torch.stack([st.losses[k].detach().float().reshape(()) for k in keys])
        torch.cuda.synchronize()
        # the static gradient tensors each segment writes, per child, in parameter order (identical on every rank)
        grads = {name: [p.grad for p in getattr(self.model, name).parameters() if p.requires_grad and p.grad is not None]
                 for name in (('discriminator', 'autoencoder') if phase == 2 else ('autoencoder',)) if hasattr(self.model, name)}
        g.update(a=ga, b=gb, c=gc, loss_vec=loss_vec, loss_keys=keys, grads=grads)
        self._restore_state(snap)
        # the graphs refresh a bank's kernel-layout weights only where the capture saw it dirty (the discriminator's: after
        # its optimizer step, not at the head of the step): bring every bank in line with the restored parameters now
        hipconvnet.refresh_stale_banks()
        torch.cuda.synchronize()
        return g


class _StepState(object):
    pass


class DurationLoss(nn.Module):
    """masked MSE between predicted and target phoneme durations (reference msmctts_trainer.py:12-36)"""

    def __init__(self, lambda_dur=1):
        super().__init__()
        self.lambda_dur = lambda_dur

    def forward(self, outputs, targets):
        dur_target = targets['dur'].float()
        lengths = targets['text_length']
        loss = F.mse_loss(outputs['duration'], dur_target, reduction='none')
        loss = loss.masked_fill(get_mask_from_lengths(lengths.to(loss.device), loss.shape[1]), 0).sum() / lengths.sum()
        return {'total_loss': self.lambda_dur * loss, 'dur_loss': loss}


class PredictorTrainer(BaseTrainer):
    """Predictor training against a frozen autoencoder (reference msmctts_trainer.py:222-295, BASELINE config #4): the
    autoencoder's ``analysis`` (eval mode, no gradient) turns the mel batch into per-stage quantised targets, the predictor
    maps text to those stages, and the loss is the per-stage embedding losses (``training_methods``, e.g. 'mse' +
    'triple_sum') plus the duration loss; gradient-norm clipping, one optimizer step on the ``predictor`` child."""

    def __init__(self, config, model, num_gpus=1, rank=0, grad_clip_thresh=1.0, eval_inteval_iters=1000,
                 training_methods=['mse'], loss_weights=[1.0], lambda_dur=1.0):
        super().__init__(config, model, num_gpus, rank)
        self.training_methods, self.loss_weights = list(training_methods), loss_weights
        self.grad_clip_thresh, self.eval_inteval_iters = grad_clip_thresh, eval_inteval_iters
        self.dur_loss = DurationLoss(lambda_dur)
        self.amp_dtype = None          # e.g. torch.bfloat16: the FFT stacks' GEMM / convolution bodies and the attention core in
        self._amp_applied = None       # that type (autocast); embeddings, losses and the optimizer stay fp32
        self.use_graphs = False        # replay the step from two hipGraphs (static shapes; HipAdamW: device-side lr / step count)
        self._graphs = None

    def _amp(self):
        if self._amp_applied is not self.amp_dtype:          # tell the HIP stacks their compute dtype
            for m in self.model.modules():
                if hasattr(m, 'hip_dtype'):
                    m.hip_dtype = self.amp_dtype or torch.float32
            self._amp_applied = self.amp_dtype
        if self.amp_dtype is None:
            return contextlib.nullcontext()
        return torch.autocast(device_type=next(self.model.parameters()).device.type, dtype=self.amp_dtype)

    def build_autoencoder(self):
        """the frozen autoencoder named by ``task.autoencoder._checkpoint`` / ``_config`` (reference :288-295)"""
        from ..tasks import load_model
        acfg = self.config.task.autoencoder
        self.autoencoder = load_model('autoencoder', acfg._checkpoint, acfg._config if hasattr(acfg, '_config') else None)
        self.autoencoder = self.autoencoder.to(next(self.model.parameters()).device)

    # the step in two halves around the gradient exchange of the data-parallel path: forward (frozen analysis, predictor,
    # losses) + backward | clip + update
    def _forward_backward(self, batch, static=False):
        """``static``: hipGraph capture -- no length is read back from the device (``MultiStagePredictor.forward(frames=)``)"""
        batch = dict(batch)
        self.autoencoder.eval()
        mel = batch.pop('mel')
        with torch.no_grad():
            qs = self.autoencoder.analysis(mel, batch.pop('mel_length').int())
        batch['feat'] = [f.float() for f in qs['quantizer_outputs']]
        batch['feat_length'] = qs['quantizer_lengths']
        # fresh dropout masks for the fused kernels of this step: the masks are hash(seed word, salt, element) and
        # only MSMCVQGAN.forward in training mode advances the word -- the autoencoder here is frozen in eval mode
        # (analysis only), so without this every predictor step would draw the SAME masks
        from ..hip import norm as hipnorm
        hipnorm.advance_seed(batch['feat'][0].device)
        with self._amp():
            output = self.model.predictor(frames=mel.shape[1], **batch) if static else self.model.predictor(**batch)
        output['feat'] = [f.float() for f in output['feat']]
        losses = {'total_loss': 0}
        emb = self.autoencoder.compute_embedding_loss(output['feat'], output['feat_length'], qs,
                                                      methods=self.training_methods, loss_weights=self.loss_weights)
        losses['total_loss'] = losses['total_loss'] + emb.pop('total_loss')
        losses.update(emb)
        dur = self.dur_loss(output, batch)
        losses['total_loss'] = losses['total_loss'] + dur.pop('total_loss')
        losses.update(dur)
        self.optimizer.zero_grad(['predictor'])
        losses['total_loss'].backward(gradient=_one_like(losses['total_loss']))
        return losses

    def _update(self, losses):
        if self.grad_clip_thresh is not None and hasattr(self.optimizer, 'clip_and_step'):
            losses['grad_norm'] = self.optimizer.clip_and_step('predictor', self.grad_clip_thresh)
        else:
            if self.grad_clip_thresh is not None:
                losses['grad_norm'] = nn.utils.clip_grad_norm_(self.model.predictor.parameters(), self.grad_clip_thresh)
            self.optimizer.step(['predictor'])

    def replays(self, iteration):
        return bool(self.use_graphs)

    def train_step(self, batch, iteration):
        if not hasattr(self, 'autoencoder'):
            self.build_autoencoder()
        if self.use_graphs:
            return self._train_step_graphed(batch)
        losses = self._forward_backward(batch)
        self._sync_grads()
        self._update(losses)
        return {'loss': {k: (v.detach() if torch.is_tensor(v) else v) for k, v in losses.items()}}

    # -- hipGraph replay (static shapes: the batch the graphs were captured on fixes text width and frame count; a batch of
    #    another shape is stepped eagerly).  Same scheme as VQGANTrainer: eager warm-up on a side stream, rolled back; two graphs
    #    sharing a pool -- forward + backward | clip + update -- with the gradient exchange between them.
    _KEYS = ('text', 'text_length', 'dur', 'mel', 'mel_length')

    def _train_step_graphed(self, batch):
        g = self._graphs
        shapes = tuple((k, tuple(batch[k].shape)) for k in self._KEYS)
        reducer = getattr(self.model, 'grad_reducer', None)
        if reducer is not None:
            # Under data parallelism every rank decides replay-or-eager from ITS OWN batch shape, so both branches below
            # exchange gradients the same way: the hooks stay off and ONE flat all-reduce of the child's gradients runs after
            # the backward pass (``allreduce_child``: the same collective, element for element, whether a rank's gradients
            # are the graphs' static tensors or this step's eager ones).  Before round 6 the eager branch relied on hooks the
            # first replay had switched off: it exchanged nothing, and a rank in it met no collective while a replaying
            # rank waited in one.
            reducer.hooks_enabled = False
        if g is not None and g['shapes'] != shapes:
            losses = self._forward_backward(batch)
            if reducer is not None:
                reducer.allreduce_child(self.model.predictor)
            self._update(losses)
            return {'loss': {k: (v.detach() if torch.is_tensor(v) else v) for k, v in losses.items()}}
        if g is None:
            g = self._graphs = self._capture(batch, shapes)
        for k in self._KEYS:
            if batch[k].data_ptr() != g['batch'][k].data_ptr():
                g['batch'][k].copy_(batch[k], non_blocking=True)
        hipconvnet.refresh_stale_banks()
        g['ab'].replay()
        if reducer is not None:
            reducer.allreduce_child(self.model.predictor, grads=g.get('grads'))
        g['c'].replay()
        hipconvnet.graphs_replayed()
        vec = g['loss_vec'].clone()
        return {'loss': {k: vec[i] for i, k in enumerate(g['loss_keys'])}}

    def _snapshot_state(self):
        opt = []
        for o in self.optimizer.optimizers.values():
            if hasattr(o, '_ensure_state'):
                for group in o.param_groups:
                    o._ensure_state(group)
            for st in o.state.values():
                opt.extend(v for v in st.values() if torch.is_tensor(v))
        return {'model': {k: v.detach().clone() for k, v in self.model.state_dict().items()},
                'opt': [(t, t.detach().clone()) for t in opt]}

    def _restore_state(self, snap):
        with torch.no_grad():
            live = self.model.state_dict()
            for k, saved in snap['model'].items():
                live[k].copy_(saved)
            for t, saved in snap['opt']:
                t.copy_(saved)
            known = set(id(t) for t, _ in snap['opt'])
            for o in self.optimizer.optimizers.values():
                for st in o.state.values():
                    for v in st.values():
                        if torch.is_tensor(v) and id(v) not in known:
                            v.zero_()

    def _capture(self, batch, shapes):
        from ..hip import graphs as hipgraphs
        dev = batch['mel'].device
        if not hipgraphs.memset_nodes_ordered(dev):
            raise RuntimeError(hipgraphs.HINT)
        snap = self._snapshot_state()
        static = {k: batch[k].clone() for k in self._KEYS}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                                   # eager warm-up (tuning, lazy buffers, optimizer state): rolled back below
                self.model.zero_grad(set_to_none=True)
                self._update(self._forward_backward(static, static=True))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        import gc as _gc
        _gc.collect()
        self.model.zero_grad(set_to_none=True)                   # gradients get (static) graph-pool storage during capture
        gab, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        import torch.distributed as dist
        mode = 'thread_local' if dist.is_available() and dist.is_initialized() else 'global'
        with torch.cuda.graph(gab, stream=side, capture_error_mode=mode):
            losses = self._forward_backward(static, static=True)
        torch.cuda.synchronize()
        prepare = getattr(self.optimizer, 'prepare', lambda names=None, tag=None: None)
        prepare(['predictor'], 'predictor')                                   # (tensor tables over the static gradients the capture allocated)
        with torch.cuda.graph(gc, pool=gab.pool(), stream=side, capture_error_mode=mode):
            self._update(losses)
            keys = [k for k, v in losses.items() if torch.is_tensor(v)]
            loss_vec = torch.stack([losses[k].detach().float().reshape(()) for k in keys])
        torch.cuda.synchronize()
        grads = [p.grad for p in self.model.predictor.parameters() if p.requires_grad and p.grad is not None]
        self._restore_state(snap)
        hipconvnet.refresh_stale_banks()
        torch.cuda.synchronize()
        return dict(ab=gab, c=gc, batch=static, shapes=shapes, loss_vec=loss_vec, loss_keys=keys, grads=grads)
