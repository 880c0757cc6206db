"""UnivNet-style discriminator: multi-resolution spectrogram (MRD) + multi-period (MPD)
(drop-in for reference msmctts/networks/hifigan/discriminator.py:15-190).

Quirks reproduced on purpose: LeakyReLU slope is 0.2 here; the MRD feature
maps the trainer sees are *post*-activation (the reference's in-place LeakyReLU aliases the stored
tensors, discriminator.py:28,71-76) while MPD's are pre-activation (:146-149).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...hip import convnet
from ...hip.convnet import ConvBank, fork_join, hip_conv, hip_conv_group, make_streams
from ...utils.audio import TorchSTFT
from ..layers import WNConv2d
from .common import get_padding

# Arithmetic: channels-last on the gfx950 implicit-GEMM kernels (csrc/conv.hip).  MPD fuses each
# leaky-ReLU into the consuming convolution's load (its feature maps are the raw outputs); MRD fuses it
# into the producing convolution's epilogue (its feature maps are the activated outputs) and the
# ReflectionPad2d into the load; feature maps are handed to the trainer as NCHW-shaped views.

LRELU_SLOPE = 0.2
# 1: under grouped execution the resolution family's launches go to a side stream (one of the library's own, hip/convnet.py
# own_streams) -- a parallel branch of the captured step, forward and backward: 17.0 -> 16.6 ms/step next to the weight
# gradients' side streams (nothing without them: alone, two chains of chip-filling launches only overlap their tails)
D_FORK = os.environ.get('MSMC_D_FORK', '1') == '1'
ACT = dict(out_slope=LRELU_SLOPE, out_masked=True)      # activation in the producer's epilogue, its backward in the consumer


class _Stage(nn.Module):
    """Keeps the reference's nn.Sequential child index of the conv ('1' for the first stage, '2' after)."""

    def __init__(self, conv, first):
        super().__init__()
        self.key = '1' if first else '2'
        self.add_module(self.key, conv)

    def forward(self, x):
        return self._modules[self.key](x)


class DiscriminatorR(nn.Module):
    def __init__(self, in_channels, hidden_channels=512):
        super().__init__()
        h = hidden_channels
        chans = [in_channels, h // 32, h // 16, h // 8, h // 4, h // 2, h, 1]
        strides = [1, 2, 1, 2, 1, 2, 1]
        self.discriminator = nn.ModuleList([
            _Stage(WNConv2d(chans[i], chans[i + 1], (3, 3), (strides[i], strides[i]), reflect_pad=1), i == 0)
            for i in range(7)])

    def hip_layers(self):
        return [st._modules[st.key].hip_layer() for st in self.discriminator]

    def forward_hip(self, bank, layers, img, dtype):
        """img channels-last [B, F, T', 2] -> (score (B, 1, F'', T''), 6 activated feature maps as NCHW-shaped views)."""
        x = img.to(dtype)
        fmaps = []
        last = len(layers) - 1
        for i, layer in enumerate(layers):
            # every map has exactly two readers, the next layer and (through that layer's tap) the feature-matching loss:
            # the leaky ReLU's backward is applied to the SUM of their gradients in the next layer's fold (ACT / in_act)
            if i == 0:
                x = hip_conv(bank, layer, x, **ACT)
            else:               # (tap: the feature-matching loss reads the alias of the map the next layer hands back)
                x, tap = hip_conv(bank, layer, x, tap=True, in_act=LRELU_SLOPE, **(ACT if i < last else {}))
                fmaps.append(tap.permute(0, 3, 1, 2))    # aliased post-activation map, see module docstring
        return x.permute(0, 3, 1, 2), fmaps


class MultiResolutionDiscriminator(nn.Module):
    def __init__(self, hop_lengths=[15, 30, 50, 120, 240, 480], hidden_channels=[128, 128, 256, 256, 512, 512],
                 domain='double', mel_scale=True, sample_rate=24000):
        super().__init__()
        self.stfts = nn.ModuleList([
            TorchSTFT(fft_size=h * 4, hop_size=h, win_size=h * 4, normalized=True, domain=domain,
                      mel_scale=mel_scale, sample_rate=sample_rate) for h in hop_lengths])
        self.domain = domain
        self.discriminators = nn.ModuleList([DiscriminatorR(2 if domain == 'double' else 1, c)
                                             for _, c in zip(hop_lengths, hidden_channels)])

    def thunks(self, bank, layers, x, dtype):
        assert self.domain == 'double', 'every shipped config uses the two-channel (mag, log-mag) image'
        wav = x.squeeze(1) if x.dim() == 3 else x
        return [lambda stft=stft, disc=disc, ls=ls: disc.forward_hip(bank, ls, stft.image_cl(wav, dtype), dtype)
                for stft, disc, ls in zip(self.stfts, self.discriminators, layers)]


class DiscriminatorP(nn.Module):
    def __init__(self, period, ch=32, max_ch=1024, kernel_size=5, stride=3, use_spectral_norm=False):
        super().__init__()
        assert not use_spectral_norm
        self.period = period
        c1, c2, c3, c4 = ch, ch * 4, min(max_ch, ch * 16), min(max_ch, ch * 32)
        pad = (get_padding(kernel_size, 1), 0)
        self.convs = nn.ModuleList([
            WNConv2d(1, c1, (kernel_size, 1), (stride, 1), pad), WNConv2d(c1, c2, (kernel_size, 1), (stride, 1), pad),
            WNConv2d(c2, c3, (kernel_size, 1), (stride, 1), pad), WNConv2d(c3, c4, (kernel_size, 1), (stride, 1), pad),
            WNConv2d(c4, c4, (5, 1), (1, 1), (2, 0))])
        self.conv_post = WNConv2d(c4, 1, (3, 1), (1, 1), (1, 0))

    def hip_layers(self):
        return [c.hip_layer() for c in self.convs] + [self.conv_post.hip_layer()]

    def forward_hip(self, bank, layers, x, dtype):
        """x (B, 1, L) -> (flat score, 5 raw feature maps as NCHW-shaped views)."""
        fmap = []
        b, c, t = x.shape
        if t % self.period != 0:
            n_pad = self.period - (t % self.period)
            x = F.pad(x, (0, n_pad), 'reflect')
            t = t + n_pad
        x = x.reshape(b, t // self.period, self.period, 1).to(dtype)        # C == 1: NCHW and NHWC coincide
        # every feature map has two consumers (the next layer and the feature-matching loss): the loss reads the alias the
        # next layer hands back (tap), so its gradient is added in that layer's data-gradient epilogue
        for i, layer in enumerate(layers[:-1]):
            if i == 0:
                x = hip_conv(bank, layer, x)
            else:
                x, tap = hip_conv(bank, layer, x, in_slope=LRELU_SLOPE, tap=True)
                fmap.append(tap.permute(0, 3, 1, 2))
        x, tap = hip_conv(bank, layers[-1], x, in_slope=LRELU_SLOPE, tap=True)
        fmap.append(tap.permute(0, 3, 1, 2))
        return torch.flatten(x, 1, -1), fmap


class MultiPeriodDiscriminator(nn.Module):
    def __init__(self, periods=[2, 3, 5, 7, 11], channels=32, max_channels=1024):
        super().__init__()
        self.discriminators = nn.ModuleList([DiscriminatorP(p, channels, max_channels) for p in periods])

    def thunks(self, bank, layers, y, dtype):
        return [lambda d=d, ls=ls: d.forward_hip(bank, ls, y, dtype) for d, ls in zip(self.discriminators, layers)]


class SpectralFronts(object):
    """The resolution discriminators' input images of one batch of waveforms (one hip/spectral.py MrdFront per hop length),
    computed once and shown to the discriminator again later: the framed-DFT front-end has no parameters, so the generator
    step's two passes (fake with gradient, real without) read rows of the images the D step already built from the very
    same waveforms -- 50 launches per step less (``VQGANTrainer._segment_a / _segment_b``)."""

    def __init__(self, fronts, r0, r1, wav=None):
        self.fronts, self.r0, self.r1, self.wav = fronts, r0, r1, wav

    def rows(self, r0, r1, wav=None):
        """the sub-batch r0 .. r1-1; ``wav``: those rows of the waveform WITH their gradient history (the generator's
        output), so that the images' gradient flows back through the saved front-end tensors"""
        return SpectralFronts(self.fronts, self.r0 + r0, self.r0 + r1, wav)

    def images(self, wavs=None):
        """``wavs``: one alias of the waveform per front (hip/spectral.py wave_fan), so that the fronts' gradients are
        summed by the fan's backward launch instead of by the autograd engine"""
        from ...hip import spectral
        if self.wav is None:
            return [f.image(self.r0, self.r1) for f in self.fronts]
        if wavs is None:
            wav = self.wav.squeeze(1) if self.wav.dim() == 3 else self.wav
            wavs = [wav] * len(self.fronts)
        return spectral.mrd_image_rows_multi([w.float() for w in wavs], self.fronts, self.r0, self.r1)


class Discriminator(nn.Module):
    def __init__(self, mrd_config, mpd_config):
        super().__init__()
        self.mrd = MultiResolutionDiscriminator(**mrd_config)
        self.mpd = MultiPeriodDiscriminator(**mpd_config)
        self.hip_dtype = torch.float32        # compute dtype of the kernels (trainer sets bfloat16 for bf16 runs)
        self._bank = None

    def _hip(self):
        """-> ((resolution bank, period bank), (resolution layers, period layers)).  TWO banks since round 6: the period
        family holds 10.3 M of the 12.9 M parameters and its backward chain (on the caller's stream) ends first, so its bank
        delivers early -- pending weight gradients, their second stages and the weight-norm backward leave on a side stream
        while the resolution family still back-propagates (0.23 ms of the discriminator step's tail ran alone on the chip
        before its optimizer could start), and under data parallelism its buckets travel under that backward (the review's
        item 7; ONE bank kept all 51 MB exposed)."""
        if self._bank is None:
            mrd = [d.hip_layers() for d in self.mrd.discriminators]
            mpd = [d.hip_layers() for d in self.mpd.discriminators]
            self._bank_r = ConvBank([l for ls in mrd for l in ls])
            self._bank_p = ConvBank([l for ls in mpd for l in ls])
            self._bank = (self._bank_r, self._bank_p)
            self._layers = (mrd, mpd)
            dev = next(self.parameters()).device
            self._streams = make_streams(dev, len(mrd) + len(mpd))
            # grouped execution: the resolution stacks' chain on ONE side stream, the period stacks' on the caller's (D_FORK)
            self._fork = convnet.own_streams(dev, 1, 'd-fork') if (convnet.GROUPED and D_FORK and dev.type == 'cuda') else []
            if self._streams:             # (MSMC_GROUPED=0: a stream per sub-discriminator, either bank's nodes on several)
                self._bank_r.streams = self._streams[:len(mrd)]
                self._bank_p.streams = self._streams[len(mrd):]
            else:
                self._bank_r.streams = self._fork
        return self._bank, self._layers

    def _prepare(self):
        (bank_r, bank_p), _ = self._hip()
        convnet.prepare_together([(bank_r, self.hip_dtype), (bank_p, self.hip_dtype)])
        bank_r.prepare(self.hip_dtype)
        bank_p.prepare(self.hip_dtype)
        return bank_r, bank_p

    def prepare_weights(self):
        """refresh the kernel-layout weight images on the calling stream if the parameters changed (forward does it too; a
        caller that is about to run two passes on two streams does it once, before the fork)"""
        self._prepare()

    def spectral_fronts(self, y):
        """the resolution discriminators' images of waveforms ``y`` (B, L) / (B, 1, L) WITHOUT gradient history, as an
        object later passes can take rows from (``forward(.., fronts=...)``)"""
        from ...hip import spectral
        wav = y.squeeze(1) if y.dim() == 3 else y
        if spectral.FRONTS_LOCKSTEP:          # every stage of the five chains in one launch (hip/spectral.py mrd_fronts)
            with torch.autocast(device_type=wav.device.type, enabled=False):
                specs = [(stft.fft_size, stft.hop_size) + tuple(stft.consts(wav.device)) for stft in self.mrd.stfts]
                return SpectralFronts(spectral.mrd_fronts(wav.detach().float(), specs, self.hip_dtype), 0, wav.shape[0])
        return SpectralFronts([stft.front(wav.detach(), self.hip_dtype) for stft in self.mrd.stfts], 0, wav.shape[0])

    def forward(self, y, fronts=None):
        """``fronts`` (optional): SpectralFronts for exactly these waveforms -- the resolution discriminators then read its
        images instead of framing and transforming ``y`` again (grouped execution only)"""
        if y.dim() == 2:
            y = y.unsqueeze(1)
        _, (mrd, mpd) = self._hip()
        bank_r, bank_p = self._prepare()
        if convnet.GROUPED:
            return self._forward_grouped(bank_r, bank_p, mrd, mpd, y, fronts)
        # the ten sub-discriminators are independent chains of small launches: one HIP stream each
        outs = fork_join(self._streams, self.mrd.thunks(bank_r, mrd, y, self.hip_dtype) +
                         self.mpd.thunks(bank_p, mpd, y, self.hip_dtype), inputs=(y,))
        return [o[0] for o in outs], [o[1] for o in outs]

    def _forward_grouped(self, bank, bank_p, mrd, mpd, y, fronts=None):
        """The sub-discriminators advance layer by layer: layer i of all six resolution (all five period) stacks is ONE
        grouped launch (hip_conv_group) -- each of them alone is a grid of tens to hundreds of workgroups."""
        dtype = self.hip_dtype
        assert self.mrd.domain == 'double', 'every shipped config uses the two-channel (mag, log-mag) image'
        wav = y.squeeze(1)
        # ONE launch makes the period stacks' inputs (cast + reflection pad to a multiple of the period) and hands the
        # resolution stacks' front-ends aliases of the waveform; its backward launch sums all ten consumers' gradients
        from ...hip import spectral
        L = wav.shape[1]
        padded = [(L + d.period - 1) // d.period * d.period for d in self.mpd.discriminators]
        fan = (len(padded) <= 8 and len(self.mrd.stfts) <= 8 and all(p <= 2 * L - 1 for p in padded)
               and wav.dtype == torch.float32 and (wav.is_cuda or spectral.lib._host_pointers_ok))
        if fan:
            wavs, copies = spectral.wave_fan(wav, len(self.mrd.stfts) if wav.requires_grad else 0, padded, dtype)
            if not wav.requires_grad:
                wavs = [wav] * len(self.mrd.stfts)
        else:
            wavs, copies = [wav] * len(self.mrd.stfts), None

        def resolution_stacks():
            if fronts is not None and fronts.fronts[0].dtype == dtype and fronts.r1 - fronts.r0 == wav.shape[0]:
                xs = fronts.images(wavs if (fan and wav.requires_grad) else None)
            else:
                xs = [stft.image_cl(w, dtype) for w, stft in zip(wavs, self.mrd.stfts)]      # (images written in the compute dtype: no cast launches)
            r_fmaps = [[] for _ in xs]
            last = len(mrd[0]) - 1
            xs = hip_conv_group(bank, [dict(layer=mrd[j][0], x=xs[j], **ACT) for j in range(len(xs))])
            for i in range(1, last + 1):
                # (tap: the feature-matching loss reads the alias of map i-1 that layer i hands back, so its gradient is added
                # in layer i's data-gradient fold -- before the leaky ReLU's derivative, which that fold applies too: in_act)
                xt = hip_conv_group(bank, [dict(layer=mrd[j][i], x=xs[j], tap=True, in_act=LRELU_SLOPE, **(ACT if i < last else {}))
                                           for j in range(len(xs))])
                for j, (x, tap) in enumerate(xt):
                    r_fmaps[j].append(tap.permute(0, 3, 1, 2))      # aliased post-activation map, see module docstring
                xs = [x for x, _ in xt]
            return [x.permute(0, 3, 1, 2) for x in xs], r_fmaps

        # the two families are independent chains of grouped launches: a parallel branch of the captured step each (their
        # backward nodes replay on the stream of their forward, so the backward pass forks the same way)
        # (a pass under no_grad does not fork: the trainer runs the generator step's D(real) pass as a side branch of its
        # own, and a fork onto a second stream from INSIDE a side branch crashed the runtime at capture_end)
        fork = self._fork if torch.is_grad_enabled() else []
        (r_scores, r_fmaps), (p_scores, p_fmaps) = fork_join(
            fork, [resolution_stacks], inputs=tuple(wavs),
            main_thunk=lambda: self._period_stacks(bank_p, mpd, y, copies, dtype))
        return r_scores + p_scores, r_fmaps + p_fmaps

    def _period_stacks(self, bank, mpd, y, copies, dtype):
        ps = []
        if copies is not None:
            for d, x in zip(self.mpd.discriminators, copies):
                ps.append(x.reshape(x.shape[0], x.shape[1] // d.period, d.period, 1))      # C == 1: NCHW and NHWC coincide
        else:
            yc = y.to(dtype)                              # ONE cast of the waveform; padding and folding in the compute dtype
            for d in self.mpd.discriminators:
                b, c, t = yc.shape
                x = yc
                if t % d.period != 0:
                    x = F.pad(x, (0, d.period - (t % d.period)), 'reflect')
                    t = x.shape[2]
                ps.append(x.reshape(b, t // d.period, d.period, 1))            # C == 1: NCHW and NHWC coincide
        p_fmaps = [[] for _ in ps]
        nl = len(mpd[0]) - 1
        # every feature map has two consumers (the next layer and the feature-matching loss): the loss reads the alias the
        # next layer hands back (tap), so its gradient is added in that layer's data-gradient epilogue
        ps = hip_conv_group(bank, [dict(layer=mpd[j][0], x=ps[j]) for j in range(len(ps))])
        for i in range(1, nl + 1):
            pt = hip_conv_group(bank, [dict(layer=mpd[j][i], x=ps[j], in_slope=LRELU_SLOPE, tap=True)
                                       for j in range(len(ps))])
            for j, (x, tap) in enumerate(pt):
                p_fmaps[j].append(tap.permute(0, 3, 1, 2))
            ps = [x for x, _ in pt]
        return [torch.flatten(x, 1, -1) for x in ps], p_fmaps
