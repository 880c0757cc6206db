"""HifiGAN residual block (drop-in for reference msmctts/networks/hifigan/common.py:8-57)."""
import torch.nn as nn

from ..layers import WNConv1d

LRELU_SLOPE = 0.1


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


class ResBlock1(nn.Module):
    """for d in dilation: x = x + conv_k,1(lrelu(conv_k,d(lrelu(x))))   (common.py:44-51)."""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.convs1 = nn.ModuleList([WNConv1d(channels, channels, kernel_size, 1, get_padding(kernel_size, d), d)
                                     for d in dilation])
        self.convs2 = nn.ModuleList([WNConv1d(channels, channels, kernel_size, 1, get_padding(kernel_size, 1), 1)
                                     for _ in dilation])

    def forward(self, x):
        raise RuntimeError('ResBlock1 is a parameter holder; Generator.forward runs it on the HIP kernels')
