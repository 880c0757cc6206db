"""HifiGAN generator (drop-in for reference msmctts/networks/hifigan/generator.py:10-64)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..layers import WNConv1d, WNConvTranspose1d
from .common import LRELU_SLOPE, ResBlock1


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

    def forward(self, mel):
        x = self.conv_pre(mel)
        for i in range(self.num_upsamples):
            x = self.ups[i](F.leaky_relu(x, LRELU_SLOPE))
            xs = None
            for j in range(self.num_kernels):
                y = self.resblocks[i * self.num_kernels + j](x)
                xs = y if xs is None else xs + y
            x = xs / self.num_kernels
        x = self.conv_post(F.leaky_relu(x))          # default slope 0.01, as the reference (generator.py:52)
        return torch.tanh(x)
