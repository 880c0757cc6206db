"""msmctts_amd -- MI355X-native MSMC-VQ-GAN training hot path.

Host-side mirror of the reference's plugin interface (``msmctts.networks`` / ``msmctts.trainers``,
hhguo/MSMC-TTS @ v2) over hand-written gfx950 kernels bound through the C ABI in
``include/msmc_hip.h``.  There is no CPU fallback: ops raise if ``libmsmc_hip.so`` is missing or a
tensor is not on the GPU.
"""
__version__ = '0.1.0'
