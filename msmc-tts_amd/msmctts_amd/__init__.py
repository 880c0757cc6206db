"""msmctts_amd -- MI355X-native MSMC-VQ-GAN training hot path.

Host-side mirror of the reference's plugin interface (``msmctts.networks`` / ``msmctts.trainers``,
hhguo/MSMC-TTS @ v2) over hand-written gfx950 kernels bound through the C ABI in
``include/msmc_hip.h``.  There is no CPU fallback: ops raise if ``libmsmc_hip.so`` is missing or a
tensor is not on the GPU.
"""
import os

# ROCm 7.2's default hipGraph path pre-builds AQL packets at instantiation ("packet capture").  On that path MEMSET
# nodes are not ordered against their neighbours on replay (tools/repro_graph_memset.py: ~half of the memsets of a
# replayed chain are still unapplied when the next node reads the buffer; kernel and memcpy nodes are fine).  PyTorch's
# multi-block reductions zero their semaphores with exactly such a node, so a train step replayed from hipGraphs got
# stale / NaN bias gradients in its stock nn.Linear layers (round-1 bench: every loss NaN).  The knob is read at the
# first HIP call, so setting it here -- before any device work of this package -- is early enough; the trainer
# verifies the behaviour with a probe before it captures (hip/graphs.py) and refuses to replay graphs otherwise.
# Cost measured on MI355X: +0.3 ms on a 35.6 ms step.
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

__version__ = '0.2.0'
