"""Weight-normalised convolution layers holding the reference's parameter names.

These modules are parameter holders: inside the HifiGAN generator, the discriminators and the quantiser's prior predictor
the arithmetic runs through ``hip_layer()`` on the gfx950 implicit-GEMM kernels (msmctts_amd/hip/convnet.py).  Their
``forward`` (stock ATen operators) is reached only by the ``norm=True`` variant of the multi-stage quantiser, which no shipped
configuration selects (networks/vqgantts/msmc_vqgan.py ``use_hip``).

The reference wraps ``torch.nn.Conv*`` in old-style ``torch.nn.utils.weight_norm`` (e.g.
hifigan/common.py:24-41, generator.py:22-35, discriminator.py:23-25,125-132, vqgantts/modules.py:209-226),
which stores ``weight_g`` (norm over all dims but 0) and ``weight_v`` next to ``bias`` -- those
``state_dict`` keys and tensor layouts are the checkpoint contract, so the
layers here own exactly those three parameters, registered in the reference's order.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..hip.convnet import ConvLayer


def _default_conv_init(weight, bias):
    nn.init.kaiming_uniform_(weight, a=math.sqrt(5))
    fan_in = weight[0].numel()
    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
    nn.init.uniform_(bias, -bound, bound)


class _WNConvNd(nn.Module):
    """Parameters: bias (C_out), weight_g (dim0, 1, ...), weight_v (conv weight layout)."""

    def __init__(self, weight_shape, n_bias):
        super().__init__()
        v = torch.empty(weight_shape)
        b = torch.empty(n_bias)
        _default_conv_init(v, b)
        self.bias = nn.Parameter(b)
        g = v.reshape(weight_shape[0], -1).norm(dim=1).reshape([weight_shape[0]] + [1] * (len(weight_shape) - 1))
        self.weight_g = nn.Parameter(g)
        self.weight_v = nn.Parameter(v)

    def weight(self):
        return torch._weight_norm(self.weight_v, self.weight_g, 0)


class WNConv1d(_WNConvNd):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__((out_channels, in_channels, kernel_size), out_channels)
        self.stride, self.padding, self.dilation = stride, padding, dilation

    def forward(self, x):
        return F.conv1d(x, self.weight(), self.bias, self.stride, self.padding, self.dilation)

    def hip_layer(self):
        return ConvLayer(self, 'conv', (1, self.weight_v.shape[2]), (1, self.stride), (1, self.dilation),
                         (0, self.padding))


class WNConvTranspose1d(_WNConvNd):
    """ConvTranspose1d: weight layout (C_in, C_out, k); weight norm is over dim 0 == C_in."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__((in_channels, out_channels, kernel_size), out_channels)
        self.stride, self.padding = stride, padding

    def forward(self, x):
        return F.conv_transpose1d(x, self.weight(), self.bias, self.stride, self.padding)

    def hip_layer(self):
        return ConvLayer(self, 'convT', (1, self.weight_v.shape[2]), (1, self.stride), (1, 1), (0, self.padding))


class WNConv2d(_WNConvNd):
    def __init__(self, in_channels, out_channels, kernel_size, stride=(1, 1), padding=(0, 0), reflect_pad=0):
        kh, kw = kernel_size
        super().__init__((out_channels, in_channels, kh, kw), out_channels)
        self.stride, self.padding, self.reflect_pad = tuple(stride), tuple(padding), reflect_pad

    def forward(self, x):
        if self.reflect_pad:
            p = self.reflect_pad
            x = F.pad(x, (p, p, p, p), mode='reflect')
        return F.conv2d(x, self.weight(), self.bias, self.stride, self.padding)

    def hip_layer(self):
        pad = (self.reflect_pad, self.reflect_pad) if self.reflect_pad else self.padding
        return ConvLayer(self, 'conv', tuple(self.weight_v.shape[2:]), self.stride, (1, 1), pad,
                         reflect=bool(self.reflect_pad))
