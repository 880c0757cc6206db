"""HifiGAN generator (drop-in for reference msmctts/networks/hifigan/generator.py:10-64).

Same constructor, parameter names and numerics; the arithmetic runs channels-last on the gfx950
implicit-GEMM kernels (csrc/conv.hip): every leaky-ReLU is fused into the consuming convolution's
load, the residual add, the sum over the three parallel ResBlocks and the ``/ num_kernels`` into
the producing convolution's epilogue, the transposed convolutions are phase-decomposed gathers and
weight norm is folded into one weight-preparation launch per forward.
"""
import torch
import torch.nn as nn

from ...hip import convnet
from ...hip.convnet import ConvBank, fork_join, hip_conv, hip_conv_group, make_streams
from ..layers import WNConv1d, WNConvTranspose1d
from .common import LRELU_SLOPE, ResBlock1

# the leaky-ReLU between the two convolutions of a ResBlock unit (reference msmctts/networks/hifigan/common.py:44-51 ResBlock1.forward) runs
# once, in the first convolution's epilogue; the second reads the activated tensor (in_act) and applies the derivative in
# its data-gradient epilogue
ACT = dict(out_slope=LRELU_SLOPE, out_masked=True)


class Generator(nn.Module):
    def __init__(self, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates, upsample_initial_channel,
                 upsample_kernel_sizes, num_mels=80):
        super().__init__()
        self.num_kernels = len(resblock_kernel_sizes)
        self.num_upsamples = len(upsample_rates)
        self.conv_pre = WNConv1d(num_mels, upsample_initial_channel, 7, 1, padding=3)
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            self.ups.append(WNConvTranspose1d(upsample_initial_channel // (2 ** i),
                                              upsample_initial_channel // (2 ** (i + 1)), k, u, padding=(k - u) // 2))
        self.resblocks = nn.ModuleList()
        ch = upsample_initial_channel
        for i in range(len(self.ups)):
            ch = upsample_initial_channel // (2 ** (i + 1))
            for k, d in zip(resblock_kernel_sizes, resblock_dilation_sizes):
                self.resblocks.append(ResBlock1(ch, k, d))
        self.conv_post = WNConv1d(ch, 1, 7, 1, padding=3)

        self.hip_dtype = torch.float32        # compute dtype of the kernels (trainer sets bfloat16 for bf16 runs)
        self._bank = None

    def _hip(self):
        if self._bank is None:
            L = {'pre': self.conv_pre.hip_layer(), 'post': self.conv_post.hip_layer(),
                 'ups': [u.hip_layer() for u in self.ups],
                 'rb': [([c.hip_layer() for c in rb.convs1], [c.hip_layer() for c in rb.convs2])
                        for rb in self.resblocks]}
            flat = [L['pre']] + L['ups'] + [c for a, b in L['rb'] for c in a + b] + [L['post']]
            self._bank, self._layers = ConvBank(flat), L
            self._streams = make_streams(self.conv_pre.weight_v.device, self.num_kernels)
            self._bank.streams = self._streams
        return self._bank, self._layers

    def forward(self, mel):
        """mel (B, C, T) -> waveform (B, 1, T * prod(upsample_rates)), float32."""
        bank, L = self._hip()
        bank.prepare(self.hip_dtype)
        nk = self.num_kernels
        x = mel.transpose(1, 2).unsqueeze(1).contiguous().to(self.hip_dtype)          # channels-last [B, 1, T, C]
        x = hip_conv(bank, L['pre'], x)
        # the mean over the nk parallel ResBlocks divides by nk in the last block's epilogue; backward, that division runs in
        # the data-gradient epilogue of the stage output's only consumer (the next up-sampling layer, ``conv_post``) instead
        # of as a tensor pass of its own (MSMC_GRAD_DIV_FUSE=0: the stock division, A/B)
        fuse_div = convnet.GRAD_DIV_FUSE
        for i in range(self.num_upsamples):
            x = hip_conv(bank, L['ups'][i], x, in_slope=LRELU_SLOPE, in_grad_div=float(nk) if (fuse_div and i > 0) else 1.0)
            def block_body(j, x=x):
                """all but the last convolution of parallel ResBlock j (independent of the other blocks)"""
                c1s, c2s = L['rb'][i * nk + j]
                y = x
                for m in range(len(c1s) - 1):
                    t, y = hip_conv(bank, c1s[m], y, in_slope=LRELU_SLOPE, tap=True, **ACT)   # (the residual edge reads the tap)
                    y = hip_conv(bank, c2s[m], t, res=y, in_act=LRELU_SLOPE)
                t, y = hip_conv(bank, c1s[-1], y, in_slope=LRELU_SLOPE, tap=True, **ACT)
                return y, t

            if convnet.GROUPED:
                # the nk parallel ResBlocks advance in lock step: one grouped launch per convolution position
                ys = [x] * nk
                nu = len(L['rb'][i * nk][0])
                # tap=True: the residual edge of a unit reads an alias of the unit's input handed back by its first
                # convolution, so the residual gradient is added in that convolution's data-gradient epilogue
                for m in range(nu - 1):
                    tt = hip_conv_group(bank, [dict(layer=L['rb'][i * nk + j][0][m], x=ys[j], in_slope=LRELU_SLOPE, tap=True, **ACT)
                                               for j in range(nk)])
                    ys = hip_conv_group(bank, [dict(layer=L['rb'][i * nk + j][1][m], x=tt[j][0], res=tt[j][1],
                                                    in_act=LRELU_SLOPE) for j in range(nk)])
                tt = hip_conv_group(bank, [dict(layer=L['rb'][i * nk + j][0][nu - 1], x=ys[j], in_slope=LRELU_SLOPE, tap=True,
                                                **ACT) for j in range(nk)])
                parts = [(t_[1], t_[0]) for t_ in tt]
            else:
                parts = fork_join(self._streams, [lambda j=j: block_body(j) for j in range(nk)], inputs=(x,))
            xs = None
            for j, (y, t) in enumerate(parts):    # block outputs, running sum over blocks and the final mean
                xs = hip_conv(bank, L['rb'][i * nk + j][1][-1], t, res=y, res2=xs, in_act=LRELU_SLOPE,
                              out_div=float(nk) if j == nk - 1 else 1.0, grad_predivided=fuse_div and j == nk - 1)
            x = xs
        x = hip_conv(bank, L['post'], x, in_slope=0.01,       # F.leaky_relu default slope (generator.py:52)
                     in_grad_div=float(nk) if (fuse_div and self.num_upsamples > 0) else 1.0)
        from ...hip import norm as hipnorm
        return hipnorm.tanh_f32(x).reshape(x.shape[0], 1, x.shape[2])       # (cast + tanh in one launch, fp32 out)
