"""ctypes front of oracle/c/vq_oracle.c (bit-exact restatement of the VQ search).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'c')
_SO = os.path.join(_DIR, 'libvq_oracle.so')
_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', _DIR])
    return _SO


def _get():
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.vq_oracle_search.restype = ctypes.c_int
        _lib.vq_oracle_search.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                                                  ctypes.c_int]
    return _lib


def search(x, embed):
    """x (N, D) float32, embed (H, d, K) float32 -> dict(ind (N,H) int64, quant (N,D), diff (N,d), dist (N,H))."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    embed = np.ascontiguousarray(embed, dtype=np.float32)
    N, D = x.shape
    H, d, K = embed.shape
    assert H * d == D
    ind = np.empty((N, H), dtype=np.int64)
    quant = np.empty((N, D), dtype=np.float32)
    diff = np.empty((N, d), dtype=np.float32)
    dist = np.empty((N, H), dtype=np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = _get().vq_oracle_search(p(x), p(embed), p(ind), p(quant), p(diff), p(dist), N, D, H, K)
    assert rc == 0
    return {'ind': ind, 'quant': quant, 'diff': diff, 'dist': dist}
