"""hipGraph replay helpers: a probe for the runtime's handling of memset nodes.

``memset_nodes_ordered(device)`` captures  fill(S, NaN) -> hipMemsetAsync(S, 0) -> R = S.clone()  a few times into one
graph and replays it: every R must be zero.  With ROCm 7.2's default "AQL packet capture" graph path that fails on
MI355X (see msmctts_amd/__init__.py, tools/repro_graph_memset.py); stock PyTorch kernels rely on such nodes (the
semaphores of multi-block reductions), so the trainer checks before it captures a train step.
"""
import ctypes

import torch

_CACHE = {}
HINT = ('hipGraph memset nodes are not ordered against their neighbours on this ROCm runtime: replayed train steps '
        'would compute stale reductions.  Set DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment BEFORE the first '
        'HIP call (importing msmctts_amd before any device work does it), or run with use_graphs = False.')


def memset_nodes_ordered(device, chain=24, replays=6, nbytes=64):
    key = torch.device(device).index or 0
    if key in _CACHE:
        return _CACHE[key]
    dev = torch.device('cuda', key)
    hip = ctypes.CDLL('libamdhip64.so')
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    bufs = [torch.empty(nbytes // 4, device=dev) for _ in range(chain)]
    pad = torch.ones(256, 256, device=dev)
    outs = []

    def work():
        del outs[:]
        y = pad
        for s in bufs:
            s.fill_(float('nan'))
            y = y @ pad * (1.0 / 256)                     # a real kernel between the fill and the memset
            rc = hip.hipMemsetAsync(s.data_ptr(), 0, nbytes, torch.cuda.current_stream(dev).cuda_stream)
            if rc != 0:
                raise RuntimeError('hipMemsetAsync failed with %d' % rc)
            outs.append(s.clone())

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        work()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        work()
    ok = True
    for _ in range(replays):
        g.replay()
        torch.cuda.synchronize(dev)
        ok = ok and all(bool((r == 0).all()) for r in outs)
    del g
    _CACHE[key] = ok
    return ok
