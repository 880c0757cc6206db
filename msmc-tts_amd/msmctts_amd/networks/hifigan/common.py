"""HifiGAN residual block (drop-in for reference msmctts/networks/hifigan/common.py:8-57).

Inside ``Generator.forward`` the three parallel blocks of a stage advance together through grouped launches that read the
blocks' parameters directly; called on its own (as a reference user may) a block runs the same kernels from a private
weight bank."""
import torch
import torch.nn as nn

from ...hip.convnet import ConvBank, hip_conv
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

        self.hip_dtype = torch.float32        # compute dtype of the kernels (float32, or bfloat16 for bf16 runs)
        self._bank = None

    def forward(self, x):
        """x (B, C, L) -> (B, C, L): the leaky-ReLU in front of a unit fused into its first convolution's load, the one between its two convolutions
        into the first one's epilogue (its derivative into the second one's data-gradient epilogue), the residual add into
        the second convolution's epilogue (and its gradient into the first one's data-gradient epilogue: tap)."""
        if self._bank is None:
            self._layers = ([c.hip_layer() for c in self.convs1], [c.hip_layer() for c in self.convs2])
            self._bank = ConvBank(self._layers[0] + self._layers[1])
        self._bank.prepare(self.hip_dtype)
        y = x.transpose(1, 2).unsqueeze(1).contiguous().to(self.hip_dtype)            # channels-last [B, 1, L, C]
        for c1, c2 in zip(*self._layers):
            t, y = hip_conv(self._bank, c1, y, in_slope=LRELU_SLOPE, tap=True, out_slope=LRELU_SLOPE, out_masked=True)
            y = hip_conv(self._bank, c2, t, res=y, in_act=LRELU_SLOPE)
        return y.squeeze(1).transpose(1, 2).to(x.dtype)
